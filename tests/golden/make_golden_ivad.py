#!/usr/bin/env python3
"""Golden fixture for the 'ivad' mask (frame VAD tiled over frequency), produced by RUNNING THE REFERENCE'S OWN CODE:
    disco_theque/sigproc_utils.py:12-55       vad_oracle_batch  (function source taken with `ast`; the module itself needs
                                              soundfile + python-acoustics)
    disco_theque/speech_enhancement/tango.py  get_mask ('ivad' branch 217-221), offline_tango
`vad_oracle_batch` calls `np.int`, which NumPy >= 1.24 removed (the reference pins numpy 1.18.1): the function is exec'd
with a numpy proxy whose only difference is `int = int`.  librosa's stft is replaced by the oracle stft as in make_golden.py.
Runs only in the build container.      python -B tests/golden/make_golden_ivad.py
"""
import ast
import copy
import os
import shutil
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
REF = '/root/reference'


class _NumpyWithInt:
    """numpy plus the removed alias np.int (what numpy 1.18 provided)."""
    int = int

    def __getattr__(self, name):
        return getattr(np, name)


def _funcs(path, names, want_assign=()):
    src = open(path).read()
    tree = ast.parse(src)
    consts, funcs = [], []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            funcs.append(ast.get_source_segment(src, node))
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in want_assign for t in node.targets):
            consts.append(ast.get_source_segment(src, node))
    return '\n'.join(consts), '\n\n'.join(funcs)


def main():
    import make_golden as mg
    scratch = tempfile.mkdtemp(prefix='disco_ref_')
    shutil.copytree(os.path.join(REF, 'disco_theque'), os.path.join(scratch, 'disco_theque'))
    sys.path.insert(0, scratch)
    from disco_theque.se_utils.internal_formulas import intern_filter
    from disco_theque.math_utils import db2lin
    from oracle import stft_oracle
    ns_v = {'np': _NumpyWithInt()}
    exec(_funcs(os.path.join(scratch, 'disco_theque/sigproc_utils.py'), {'vad_oracle_batch'})[1], ns_v)
    vad_oracle_batch = ns_v['vad_oracle_batch']
    ns_m = {'np': np, 'sys': sys, 'db2lin': db2lin}
    exec(_funcs(os.path.join(scratch, 'disco_theque/dnn/utils.py'), {'tf_mask'})[1], ns_m)
    lb = types.SimpleNamespace(core=types.SimpleNamespace(
        stft=lambda x, n_fft, hop_length, center: stft_oracle.stft(x, n_fft, hop_length, 'reflect')))
    ns = {'np': np, 'copy': copy, 'lb': lb, 'tf_mask': ns_m['tf_mask'], 'intern_filter': intern_filter,
          'vad_oracle_batch': vad_oracle_batch, 'prepare_data': None}
    consts, funcs = _funcs(os.path.join(scratch, 'disco_theque/speech_enhancement/tango.py'),
                           {'concatenate_signals', 'get_z_for_mask', 'get_mask', 'reshape_mask', 'offline_tango'},
                           want_assign={'N_FFT', 'N_HOP', 'nb_ch', 'nb_nodes', 'ref_mics', 'WIN_LEN', 'PRED_FRAME', 'MASK_Z'})
    exec(consts, ns)
    exec(funcs, ns)
    rng = np.random.default_rng(424242)
    out = {}
    # ---- the VAD itself on signals with silences, bursts, a DC offset and a length that is not a multiple of the hop
    sigs = []
    for i, L in enumerate((6000, 5121, 4096)):
        x = rng.standard_normal(L) * (rng.random(L) > 0.2)
        env = np.ones(L)
        env[:L // 6] = 0.0
        env[L // 2:L // 2 + 700] = 0.01
        env[-300:] = 0.0 if i != 1 else 1.0
        x = (x * env + 0.05 * i).astype(np.float32)
        sigs.append(x)
        out[f'vad_x{i}'] = x
        out[f'vad_o{i}'] = vad_oracle_batch(x, win_len=512, win_hop=256)
    out['n_vad'] = np.array(len(sigs))
    # ---- whole path with 'ivad' masks on the k2m2 toy scene
    K, Mk, L = 2, [2, 2], 4096
    y, s, n = mg._toy_scene(rng, K, Mk, L)
    res = ns['offline_tango'](y, s, n, vads=['ivad', 'ivad'], mods=[None, None], mask_for_z='local')
    names = ['yf', 'sf', 'nf', 'z_y', 'z_s', 'z_n', 'zn', 'masks_z', 'mask_w']
    out.update(K=np.array(K), L=np.array(L))
    for k in range(K):
        out[f'y{k}'], out[f's{k}'], out[f'n{k}'] = y[k], s[k], n[k]
        for nm, arr in zip(names, res):
            out[f'{nm}{k}'] = np.asarray(arr[k])
    np.savez_compressed(os.path.join(HERE, 'ivad_ref.npz'), **out)
    shutil.rmtree(scratch, ignore_errors=True)
    print('wrote ivad_ref.npz; active fraction', [float(out[f'vad_o{i}'].mean()) for i in range(3)],
          'mask mean', float(out['masks_z0'].mean()))


if __name__ == '__main__':
    main()
