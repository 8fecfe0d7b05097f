"""CRNN mask estimator of DISCO, PyTorch-ROCm -- the one part of the path the north star keeps in
PyTorch.  Architecture and `state_dict` key names follow the reference so that reference-trained checkpoints load:
    disco_theque/dnn/models/crnn.py:9-63        CRNN  (3 x [Conv2d 3x3 -> BatchNorm2d -> MaxPool (1,4)] -> GRU-256 -> Linear-257 -> sigmoid)
    disco_theque/dnn/models/nn_structures.py    CNN2d / RNN / FF bricks   (key prefixes `cnn.model.*`, `rnn.model.0.rnn_layer.*`, `ff.layers.0.*`)
    disco_theque/speech_enhancement/tango.py:114-139  load_models: CRNN((n_ch, 21, 257), (32,64,64), 3x3, pool (1,4), GRU [256], FF 257, padding (0,1))
    disco_theque/speech_enhancement/utils.py:69-138   prepare_data: clip |STFT| to [1e-6, 1e3], zero-pad 10+10 frames, 21-frame windows, hop 1
    disco_theque/speech_enhancement/tango.py:228-240  reshape_mask('mid'): output frame 7 of the 15 the network returns

Two evaluation paths with identical results:
  * `forward(windows)`      -- the reference's: one (n_ch, 21, 257) window per output frame.
  * `predict_masks(mag)`    -- what the engine uses: the convolutional stack has no padding along time and pools along
    frequency only, so it is run ONCE over the whole zero-padded sequence and the 15-frame feature windows are slices of
    that; the reference's `.view` (which re-interprets the (64, 15, 4) block as (15, 256) WITHOUT a transpose,
    crnn.py:59) is reproduced on the slices; only the 8 GRU steps that reach output frame 7 are run.
"""
import numpy as np
import torch
from torch import nn

STFT_MIN, STFT_MAX = 1e-6, 1e3          # speech_enhancement/utils.py:7
WIN_LEN = 21                            # tango.py:34
PRED_FRAME = 'mid'                      # tango.py:35


class _Seq(nn.Module):
    """Holder that reproduces the reference's `<brick>.model = nn.Sequential(...)` key layout."""

    def __init__(self, *mods):
        super().__init__()
        self.model = nn.Sequential(*mods)

    def forward(self, x):
        return self.model(x)


class _RnnLayer(nn.Module):
    """`rnn.model.<i>.rnn_layer.*` (nn_structures.py RNNSingle): returns the sequence output only."""

    def __init__(self, input_size, hidden):
        super().__init__()
        self.rnn_layer = nn.GRU(input_size=input_size, hidden_size=hidden, num_layers=1, batch_first=True)

    def forward(self, x):
        return self.rnn_layer(x)[0]


class _FF(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.layers = nn.ModuleList([nn.Linear(n_in, n_out)])

    def forward(self, x):
        return torch.sigmoid(self.layers[0](x))


class CRNN(nn.Module):
    def __init__(self, n_ch=1, win_len=WIN_LEN, n_freq=257, cnn_filters=(32, 64, 64), rnn_units=256):
        super().__init__()
        self.input_shape = (n_ch, win_len, n_freq)
        chans = [n_ch, *cnn_filters]
        mods = []
        f = n_freq
        for i in range(len(cnn_filters)):
            mods += [nn.Conv2d(chans[i], chans[i + 1], kernel_size=3, stride=1, padding=(0, 1)),
                     nn.BatchNorm2d(chans[i + 1]), nn.MaxPool2d((1, 4))]
            f = f // 4
        self.cnn = _Seq(*mods)
        self.x_out = win_len - 2 * len(cnn_filters)                 # 15 frames survive the three unpadded 3x3 convolutions
        self.y_out = f                                              # 4 frequency cells after three (1,4) poolings
        self.rnn = _Seq(_RnnLayer(chans[-1] * self.y_out, rnn_units))
        self.ff = _FF(rnn_units, n_freq)

    # ---- the reference's evaluation (crnn.py:55-63)
    def forward(self, inp):
        if inp.dim() == 3:
            inp = inp.view(inp.size(0), 1, inp.size(1), inp.size(2))
        x = self.cnn(inp)
        x = x.view(x.size(0), x.size(2), x.size(1) * x.size(-1))    # NB: a re-interpretation, not a transpose (crnn.py:59)
        x = self.rnn(x)
        return self.ff(x.squeeze())

    def mid_frame(self):
        """Index of the output frame reshape_mask('mid') selects (tango.py:232-234)."""
        return int(np.floor(self.x_out / 2))

    # ---- sequence evaluation
    @torch.no_grad()
    def predict_masks(self, mag, chunk=256):
        """mag: (B, n_ch, T, F) magnitudes (un-clipped |STFT| of the node's reference mic, then |z| of the other nodes)
        -> masks (B, T, F), equal to reshape_mask(model(prepare_data(...)), 'mid') of the reference for every item."""
        B, C, T, F = mag.shape
        pad = WIN_LEN // 2                                          # get_frames_to_pad('mid'): (10, 10)
        x = torch.clamp(mag, STFT_MIN, STFT_MAX)                    # normalization(norm_type=None)
        x = torch.nn.functional.pad(x, (0, 0, pad, pad))            # zeros AFTER clipping, as prepare_data does
        feat = self.cnn(x)                                          # (B, 64, T + 20 - 6, 4)
        Cc, W, Fy = feat.shape[1], self.x_out, self.y_out
        steps = self.mid_frame() + 1                                # GRU steps needed to reach the selected output frame
        gru = self.rnn.model[0].rnn_layer
        out = torch.empty((B, T, F), dtype=mag.dtype, device=mag.device)
        for b0 in range(0, B, chunk):
            fb = feat[b0:b0 + chunk]
            nb = fb.shape[0]
            win = fb.unfold(2, W, 1)                                # (nb, 64, T, 4, 15): window i = frames i .. i+14
            win = win.permute(0, 2, 1, 4, 3).reshape(nb * T, Cc * W * Fy)        # each row = the (64, 15, 4) block, C-order
            seq = win.view(nb * T, W, Cc * Fy)[:, :steps, :]        # the reference's .view, truncated to the needed steps
            h = gru(seq.contiguous())[0][:, -1, :]                  # hidden state after step `mid`
            out[b0:b0 + chunk] = self.ff(h).view(nb, T, F)
        return out


def build_crnn(n_ch=1, device=None, state_dict=None):
    """CRNN with the constructor arguments of tango.py:124-129; optionally loads a reference checkpoint's
    `model_state_dict` (train.py:151-156)."""
    model = CRNN(n_ch=n_ch)
    if state_dict is not None:
        model.load_state_dict(state_dict)
    if device is not None:
        model = model.to(device)
    return model.eval()


def get_z_for_mask(z_s, z_n, k, nb_nodes, z_sigs='zs_hat'):
    """tango.py:158-186 -- which compressed signals feed the step-2 network of node k.  z_s, z_n: (K, ...) arrays/tensors."""
    if z_sigs in ('zs_hat', 'zn_hat'):
        z_in = z_s if z_sigs == 'zs_hat' else z_n
        idx = [j for j in range(nb_nodes) if j != k]
        return z_in[idx]
    cat = torch.cat if torch.is_tensor(z_s) else np.concatenate
    z_in = cat((z_s, z_n), 0)
    n = z_in.shape[0]
    order = [i // 2 if i % 2 == 0 else int(0.5 * (n - 1 + i)) for i in range(n)]      # interleave zs_j, zn_j as in training
    z_out = z_in[order]
    keep = [i for i in range(2 * nb_nodes) if i not in (2 * k, 2 * k + 1)]
    return z_out[keep]
