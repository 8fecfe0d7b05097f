"""TEST INFRASTRUCTURE (CPU oracle) -- only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

Online / adaptive mode of the two-step MWF.  The reference ships the primitive and no loop:
    disco_theque/se_utils/internal_formulas.py:84-103   spatial_correlation_matrix  R <- lambda R + M (1-lambda) x x^H
    disco_theque/se_utils/internal_formulas.py:56-73    intern_filter(..., 'gevd', rank=1)
The recursion restated here is the one those two define when driven frame by frame (the batch means of tango.py:357-364
replaced by the exponential smoothing; the mask multiplies the outer product ONCE, as the primitive's docstring says --
not squared as in the batch path):

    Rss_t = lambda Rss_{t-1} + (1-lambda)      m_t  v_t v_t^H          Rss_-1 = 0
    Rnn_t = lambda Rnn_{t-1} + (1-lambda) (1 - m_t) v_t v_t^H          Rnn_-1 = init_diag * I
    w_t   = intern_filter(Rss_t, Rnn_t, mu, 'gevd', 1)   when t % update_every == 0, else w_{t-1}
    out_t = w_t^H v_t

Pinned by tests/golden/online_ref.npz, which tests/golden/make_golden_online.py produced by calling the reference's own
two functions in this recursion.  Everything is float64 (the HIP kernel keeps the smoothed matrices in float32 and solves
in float64; the forgetting factor stops rounding from accumulating).
"""
import numpy as np

from . import mwf_oracle, stft_oracle


def online_mwf(V, mask, lambda_cor=0.95, mu=1.0, update_every=1, init_diag=1e-3):
    """V (P, F, T) complex, mask (F, T) -> out (F, T) c128, w (F, T, P) c128 (the filter in force at every frame)."""
    V = np.asarray(V).astype(np.complex128)
    mask = np.asarray(mask).astype(np.float64)
    P, F, T = V.shape
    Rss = np.zeros((F, P, P), np.complex128)
    Rnn = np.tile(init_diag * np.eye(P, dtype=np.complex128), (F, 1, 1))
    w = np.zeros((F, P), np.complex128)
    out = np.zeros((F, T), np.complex128)
    w_all = np.zeros((F, T, P), np.complex128)
    for t in range(T):
        v = V[:, :, t].T                                                # (F, P)
        vv = v[:, :, None] * np.conjugate(v)[:, None, :]                # np.outer(x, conj(x).T) per bin
        m = mask[:, t][:, None, None]
        Rss = lambda_cor * Rss + m * (1 - lambda_cor) * vv              # internal_formulas.py:102, M = m
        Rnn = lambda_cor * Rnn + (1 - m) * (1 - lambda_cor) * vv        # internal_formulas.py:102, M = 1 - m
        if t % update_every == 0:
            w = mwf_oracle.gevd_mwf_r1_hermitian(Rss, Rnn, mu)[0]       # == intern_filter 'gevd' rank 1 (pinned)
        w_all[:, t] = w
        out[:, t] = np.einsum('fp,fp->f', np.conjugate(w), v)
    return out, w_all


def online_tango(y, s, n, n_fft=512, hop=256, pad_mode='reflect', mask_type='irm1', lambda_cor=0.95, mu=1.0,
                 update_every=1, init_diag=1e-3):
    """Two-step online MWF of one room.  y, s, n: (K, M, L) float32.  Step 1 runs `online_mwf` on every node's own
    channels -> z_k (causal, frame by frame); step 2 runs it on [Y_k ; z_j (j<k) ; z_j (j>k)] (tango.py:142-155 order,
    the node's own mask on every row = mask_for_z 'local', tango.py:36) -> yf_k -> iSTFT.
    Returns dict(out (K, L) f32, z (K, F, T), yf (K, F, T), masks (K, F, T))."""
    y, s, n = (np.asarray(a, dtype=np.float32) for a in (y, s, n))
    K, M, L = y.shape
    Y = np.stack([np.stack([stft_oracle.stft(y[k, m], n_fft, hop, pad_mode) for m in range(M)]) for k in range(K)])
    masks = []
    for k in range(K):
        S = stft_oracle.stft(s[k, 0], n_fft, hop, pad_mode)
        N = stft_oracle.stft(n[k, 0], n_fft, hop, pad_mode)
        masks.append(mwf_oracle.tf_mask(S, N, mask_type))
    masks = np.stack(masks)
    z = np.stack([online_mwf(Y[k], masks[k], lambda_cor, mu, update_every, init_diag)[0] for k in range(K)])
    z32 = z.astype(np.complex64)                        # the exchanged signal is complex64 (as the batch path's z)
    yf = []
    for k in range(K):
        rows = [Y[k]] + [z32[j][None] for j in range(K) if j < k] + [z32[j][None] for j in range(K) if j > k]
        yf.append(online_mwf(np.concatenate(rows, 0), masks[k], lambda_cor, mu, update_every, init_diag)[0])
    yf = np.stack(yf)
    out = np.stack([stft_oracle.istft(yf[k].astype(np.complex64), L, n_fft, hop) for k in range(K)])
    return {'out': out, 'z': z, 'yf': yf, 'masks': masks}
