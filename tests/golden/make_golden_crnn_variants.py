#!/usr/bin/env python3
"""More of the reference's mask-estimation front end, again by RUNNING THE REFERENCE'S OWN CODE (same machinery as
make_golden_crnn.py, whose 'sc' / 'mc' weights are re-used from crnn_ref.npz so that no second copy of them is stored):
  * prepare_data(..., frame_to_pred='last') + reshape_mask(..., 'last')      (speech_enhancement/utils.py:13-33, tango.py:228-231)
  * normalization(norm_type = 'scale_to_unit_norm' | 'scale_to_1' | 'center_and_scale')      (utils.py:36-66)
Writes tests/golden/crnn_variants_ref.npz (build container only)."""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden_crnn as mc  # noqa: E402


def main():
    scratch = tempfile.mkdtemp(prefix='disco_ref_')
    try:
        shutil.copytree(os.path.join(mc.REF, 'disco_theque'), os.path.join(scratch, 'disco_theque'))
        sys.path.insert(0, scratch)
        from disco_theque.dnn.models.nn_structures import CNN2d, FF, RNN
        from torch import nn
        ns = {'np': np, 'nn': nn, 'torch': torch, 'CNN2d': CNN2d, 'RNN': RNN, 'FF': FF}
        exec(mc.seg(os.path.join(scratch, 'disco_theque/dnn/utils.py'), {'get_loss_frames'}), ns)
        exec(mc.seg(os.path.join(scratch, 'disco_theque/dnn/models/crnn.py'), {'CRNN'}), ns)
        nsu = {'np': np, 'torch': torch, 'lb': None}
        exec(mc.seg(os.path.join(scratch, 'disco_theque/speech_enhancement/utils.py'),
                    {'get_frames_to_pad', 'normalization', 'prepare_data'},
                    assigns={'stft_min', 'stft_max', 'fs', 'n_hop', 'frames_lost'}), nsu)
        prepare_data = nsu['prepare_data']
        nst = {'np': np, 'nb_nodes': 4}
        exec(mc.seg(os.path.join(scratch, 'disco_theque/speech_enhancement/tango.py'), {'reshape_mask'}), nst)
        reshape_mask = nst['reshape_mask']
        gold = np.load(os.path.join(HERE, 'crnn_ref.npz'))
        d = {}
        for tag, n_ch in (('sc', 1), ('mc', 4)):
            model = ns['CRNN']((n_ch, 21, 257), (32, 64, 64), (3, 3, 3), (1, 1, 1), [(1, 4), (1, 4), (1, 4)], (None, None, None),
                               [256], 'GRU', 257, conv_padding=[(0, 1), (0, 1), (0, 1)])
            model.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f'{tag}_sd_')})
            model.eval()
            Y = gold[f'{tag}_Y']
            Z = list(gold[f'{tag}_Z']) if n_ch > 1 else None
            lost = int(21 - model.get_loss_frames('last')[-1][-1])
            for ftp, nt in (('last', None), ('mid', 'scale_to_unit_norm'), ('mid', 'scale_to_1'), ('last', 'center_and_scale')):
                x_in = prepare_data(Y, True, z_data=Z, win_len=21, win_hop=1, frame_to_pred=ftp, norm_type=nt, frames_lost=lost)
                with torch.no_grad():
                    m_stack = model(x_in.cpu()).detach().numpy()
                d[f'{tag}_{ftp}_{nt}'] = reshape_mask(m_stack, ftp)
        np.savez_compressed(os.path.join(HERE, 'crnn_variants_ref.npz'), **d)
        print('wrote crnn_variants_ref.npz', {k: v.shape for k, v in d.items()})
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
