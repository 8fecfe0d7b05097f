#!/usr/bin/env python3
"""Golden fixtures for the CRNN mask estimator, produced by RUNNING THE REFERENCE'S OWN CODE:
  * disco_theque/dnn/models/nn_structures.py           imported from a scratch copy (it imports cleanly)
  * disco_theque/dnn/models/crnn.py  class CRNN         exec'd from its source segment (the module itself cannot be
                                                        imported: circular import with dnn/utils.py)
  * disco_theque/dnn/utils.py  get_loss_frames          exec'd from its source segment
  * disco_theque/speech_enhancement/utils.py  prepare_data, normalization, get_frames_to_pad, constants  (exec'd; the
                                                        module imports librosa and a non-existent move_to_device)
  * disco_theque/speech_enhancement/tango.py  reshape_mask, get_z_for_mask                    (exec'd)
with the constructor arguments of tango.py:124-129 (load_models).  Weights are random (seeded): no trained
checkpoint ships with the reference.  Writes tests/golden/crnn_ref.npz (build container only)."""
import ast
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.dont_write_bytecode = True


def seg(path, names, assigns=()):
    src = open(path).read()
    out = []
    for node in ast.parse(src).body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            out.append(ast.get_source_segment(src, node))
        if isinstance(node, ast.Assign):
            names_ = [n.id for t in node.targets for n in ast.walk(t) if isinstance(n, ast.Name)]
            if any(n in assigns for n in names_):
                out.insert(0, ast.get_source_segment(src, node))
    return '\n\n'.join(out)


def main():
    scratch = tempfile.mkdtemp(prefix='disco_ref_')
    try:
        shutil.copytree(os.path.join(REF, 'disco_theque'), os.path.join(scratch, 'disco_theque'))
        sys.path.insert(0, scratch)
        from disco_theque.dnn.models.nn_structures import CNN2d, FF, RNN          # real reference code
        from torch import nn
        ns = {'np': np, 'nn': nn, 'torch': torch, 'CNN2d': CNN2d, 'RNN': RNN, 'FF': FF}
        exec(seg(os.path.join(scratch, 'disco_theque/dnn/utils.py'), {'get_loss_frames'}), ns)
        exec(seg(os.path.join(scratch, 'disco_theque/dnn/models/crnn.py'), {'CRNN'}), ns)
        CRNN = ns['CRNN']
        nsu = {'np': np, 'torch': torch, 'lb': None}
        exec(seg(os.path.join(scratch, 'disco_theque/speech_enhancement/utils.py'),
                 {'get_frames_to_pad', 'normalization', 'prepare_data'},
                 assigns={'stft_min', 'stft_max', 'fs', 'n_hop', 'frames_lost'}), nsu)
        prepare_data = nsu['prepare_data']
        nst = {'np': np, 'nb_nodes': 4}
        exec(seg(os.path.join(scratch, 'disco_theque/speech_enhancement/tango.py'), {'reshape_mask', 'get_z_for_mask'}), nst)
        reshape_mask, get_z_for_mask = nst['reshape_mask'], nst['get_z_for_mask']

        d = {}
        rng = np.random.default_rng(4242)
        for tag, n_ch in (('sc', 1), ('mc', 4)):           # single-channel (step 1) and 1 + (K-1) channels (step 2, K = 4)
            torch.manual_seed(100 + n_ch)
            model = CRNN((n_ch, 21, 257), (32, 64, 64), (3, 3, 3), (1, 1, 1), [(1, 4), (1, 4), (1, 4)], (None, None, None),
                         [256], 'GRU', 257, conv_padding=[(0, 1), (0, 1), (0, 1)])            # tango.py:124-129
            # non-trivial batch-norm statistics, then eval mode as in get_mask (tango.py:210)
            for m in model.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.running_mean.normal_(0, 0.2)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.data.uniform_(0.5, 1.5)
                    m.bias.data.normal_(0, 0.1)
            model.eval()
            T = 37
            Y = (rng.standard_normal((257, T)) + 1j * rng.standard_normal((257, T))).astype(np.complex64) * 0.5
            Z = None
            if n_ch > 1:
                Z = [(rng.standard_normal((257, T)) + 1j * rng.standard_normal((257, T))).astype(np.complex64) * 0.3
                     for _ in range(n_ch - 1)]
            lost = int(21 - model.get_loss_frames('last')[-1][-1])
            x_in = prepare_data(Y, True, z_data=Z, win_len=21, win_hop=1, frame_to_pred='mid', frames_lost=lost)   # (T, n_ch, 21, 257)
            with torch.no_grad():
                m_stack = model(x_in.cpu()).detach().numpy()                                  # (T, 15, 257)
            mask = reshape_mask(m_stack, 'mid')                                               # (257, T)
            for k, v in model.state_dict().items():
                d[f'{tag}_sd_{k}'] = v.numpy()
            d[f'{tag}_Y'] = Y
            if Z is not None:
                d[f'{tag}_Z'] = np.array(Z)
            d[f'{tag}_x_in'] = x_in.cpu().numpy()[:5]                       # first windows only (size)
            d[f'{tag}_m_stack'] = m_stack[:5]
            d[f'{tag}_mask'] = mask
            d[f'{tag}_lost'] = np.array(lost)
        # get_z_for_mask ordering (tango.py:158-186) on labelled arrays
        zs = np.arange(4)[:, None, None] * np.ones((4, 2, 3))
        zn = -np.arange(1, 5)[:, None, None] * np.ones((4, 2, 3))
        for k in range(4):
            d[f'zfm_zs_hat_{k}'] = get_z_for_mask(zs, zn, k, 4, 'zs_hat')
            d[f'zfm_both_{k}'] = get_z_for_mask(zs, zn, k, 4, ['zs_hat', 'zn_hat'])
        np.savez_compressed(os.path.join(HERE, 'crnn_ref.npz'), **d)
        print('wrote crnn_ref.npz', {k: v.shape for k, v in d.items() if not k.startswith(('sc_sd', 'mc_sd'))})
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
