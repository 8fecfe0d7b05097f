#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by RUNNING THE REFERENCE'S OWN CODE.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite only read
the committed .npz files.  Usage:   python -B tests/golden/make_golden.py

What is executed from the reference (loaded at run time, never copied into this repository):
  * disco_theque/se_utils/internal_formulas.py  -- imported as a module from a scratch copy under /tmp
    (never import from /root/reference in place, it would write __pycache__ there).
  * disco_theque/dnn/utils.py `tf_mask`          -- the module cannot be imported (circular import with
    dnn/models/crnn.py), so the function's source segment is taken from the file with `ast` and exec'd.
  * disco_theque/speech_enhancement/tango.py `offline_tango`, `concatenate_signals`, `get_mask`,
    `get_z_for_mask`, `reshape_mask` and the module constants (N_FFT, N_HOP, ref_mics, ...) -- the module
    cannot be imported (librosa, soundfile, mir_eval, pystoi, ipdb and a missing dnn/models/heymann.py),
    so the function definitions are taken with `ast` and exec'd in a namespace where the ONLY substituted
    piece is `lb.core.stft` -> oracle.stft_oracle.stft (librosa is third-party and absent).
The outputs therefore pin: intern_filter, tf_mask, and the full two-step loop nest (ordering, conjugation,
dtype flow, concatenation order) of the reference itself.
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
sys.dont_write_bytecode = True

REF = '/root/reference'


def _load_reference():
    scratch = tempfile.mkdtemp(prefix='disco_ref_')
    shutil.copytree(os.path.join(REF, 'disco_theque'), os.path.join(scratch, 'disco_theque'))
    sys.path.insert(0, scratch)
    from disco_theque.se_utils.internal_formulas import intern_filter          # real reference code
    from disco_theque.math_utils import db2lin                                  # real reference code
    from oracle import stft_oracle

    def seg(path, names, want_assign=()):
        src = open(path).read()
        tree = ast.parse(src)
        consts, funcs = [], []
        for node in tree.body:
            if isinstance(node, ast.FunctionDef) and node.name in names:
                funcs.append(ast.get_source_segment(src, node))
            if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in want_assign for t in node.targets):
                consts.append(ast.get_source_segment(src, node))
        return '\n'.join(consts), '\n\n'.join(funcs)

    ns_mask = {'np': np, 'sys': sys, 'db2lin': db2lin}
    exec(seg(os.path.join(scratch, 'disco_theque/dnn/utils.py'), {'tf_mask'})[1], ns_mask)
    tf_mask = ns_mask['tf_mask']

    lb = types.SimpleNamespace(core=types.SimpleNamespace(
        stft=lambda x, n_fft, hop_length, center: stft_oracle.stft(x, n_fft, hop_length, 'reflect')))
    ns = {'np': np, 'copy': copy, 'lb': lb, 'tf_mask': tf_mask, 'intern_filter': intern_filter,
          'vad_oracle_batch': None, 'prepare_data': None}
    consts, funcs = seg(os.path.join(scratch, 'disco_theque/speech_enhancement/tango.py'),
               {'concatenate_signals', 'get_z_for_mask', 'get_mask', 'reshape_mask', 'offline_tango'},
               want_assign={'N_FFT', 'N_HOP', 'nb_ch', 'nb_nodes', 'ref_mics', 'WIN_LEN', 'PRED_FRAME', 'MASK_Z'})
    # constants first (they are default-argument values of the functions)
    exec(consts, ns)
    exec(funcs, ns)
    return intern_filter, tf_mask, ns['offline_tango'], scratch


def _toy_scene(rng, K, Mk, L):
    """Small, well-conditioned multichannel scene: one target and one noise source through short random
    FIRs plus weak sensor noise (so every covariance is full rank even with few frames)."""
    y, s, n = [], [], []
    src_s = rng.standard_normal(L) * np.concatenate([np.zeros(L // 8), np.ones(L - L // 8)])
    src_n = rng.standard_normal(L)
    for k in range(K):
        sk, nk = [], []
        for c in range(Mk[k]):
            hs = rng.standard_normal(24) * np.exp(-np.arange(24) / 6.0)
            hn = rng.standard_normal(24) * np.exp(-np.arange(24) / 6.0)
            sk.append(0.3 * np.convolve(src_s, hs)[:L] + 0.01 * rng.standard_normal(L) * (np.arange(L) >= L // 8))
            nk.append(0.2 * np.convolve(src_n, hn)[:L] + 0.02 * rng.standard_normal(L))
        sk = np.array(sk, np.float32)
        nk = np.array(nk, np.float32)
        s.append(sk)
        n.append(nk)
        y.append(sk + nk)
    return y, s, n


def main():
    intern_filter, tf_mask, offline_tango, scratch = _load_reference()
    try:
        rng = np.random.default_rng(20260921)

        # ---- intern_filter: 'gevd' rank 1 (live branch), 'r1-mwf', 'mwf'
        cases = {}
        i = 0
        for P in (2, 4, 7, 15):
            for T in (3 * P, 40 * P):
                for dt in (np.complex64, np.complex128):
                    a = rng.standard_normal((P, 1)) + 1j * rng.standard_normal((P, 1))
                    X = a @ (rng.standard_normal((1, T)) + 1j * rng.standard_normal((1, T))) \
                        + 0.1 * (rng.standard_normal((P, T)) + 1j * rng.standard_normal((P, T)))
                    Nn = rng.standard_normal((P, T)) + 1j * rng.standard_normal((P, T))
                    Rxx = (X @ X.conj().T / T).astype(dt)
                    Rnn = (Nn @ Nn.conj().T / T).astype(dt)
                    for typ in ('gevd', 'r1-mwf', 'mwf'):
                        w, (t1, si) = intern_filter(Rxx, Rnn, mu=1, type=typ, rank=1)
                        cases[f'c{i}_Rxx'] = Rxx
                        cases[f'c{i}_Rnn'] = Rnn
                        cases[f'c{i}_type'] = np.array(typ)
                        cases[f'c{i}_w'] = np.asarray(w)
                        cases[f'c{i}_t1'] = np.asarray(t1)
                        cases[f'c{i}_sort'] = np.asarray(-1 if si is None else si)
                        i += 1
        cases['n_cases'] = np.array(i)
        np.savez_compressed(os.path.join(HERE, 'intern_filter_ref.npz'), **cases)
        print('intern_filter cases:', i)

        # ---- tf_mask
        S = (rng.standard_normal((33, 12)) + 1j * rng.standard_normal((33, 12))).astype(np.complex64)
        N = (rng.standard_normal((33, 12)) + 1j * rng.standard_normal((33, 12))).astype(np.complex64)
        N[3, 4] = 0
        S[5, 6] = 0
        md = {'S': S, 'N': N}
        for typ in ('irm1', 'irm2', 'ibm1', 'iam1', 'iam2'):
            md[typ] = tf_mask(S, N, type=typ)
        np.savez_compressed(os.path.join(HERE, 'tf_mask_ref.npz'), **md)

        # ---- offline_tango (real reference loop nest; librosa stft replaced by the oracle stft)
        scenes = [('k2m2', 2, [2, 2], 4096), ('k3ragged', 3, [3, 2, 2], 5120), ('k4m4', 4, [4, 4, 4, 4], 6144)]
        for name, K, Mk, L in scenes:
            y, s, n = _toy_scene(rng, K, Mk, L)
            if name == 'k2m2':
                # the other mask_for_z modes (tango.py:396-429), yf / sf / nf only, on the smallest scene
                dm = {'K': np.array(K), 'L': np.array(L)}
                for k in range(K):
                    dm[f'y{k}'], dm[f's{k}'], dm[f'n{k}'] = y[k], s[k], n[k]
                for mfz in ('distant', 'compressed', 'use_oracle_refs', 'use_oracle_zs', 'previous'):
                    res = offline_tango(y, s, n, vads=['irm1', 'irm1'], mods=[None, None], mask_for_z=mfz)
                    for k in range(K):
                        for nm, arr in zip(['yf', 'sf', 'nf'], res[:3]):
                            dm[f'{mfz}_{nm}{k}'] = np.asarray(arr[k])
                np.savez_compressed(os.path.join(HERE, 'tango_ref_modes_k2m2.npz'), **dm)
                print('mask_for_z modes done')
            for mfz in ('local', None):
                if mfz is None:
                    # tango.py:343 does `'use_oracle_' in mask_for_z`, a TypeError for None as shipped; the
                    # None mode is therefore not runnable in the reference and is not pinned.
                    continue
                res = offline_tango(y, s, n, vads=['irm1', 'irm1'], mods=[None, None], mask_for_z=mfz)
                names = ['yf', 'sf', 'nf', 'z_y', 'z_s', 'z_n', 'zn', 'masks_z', 'mask_w']
                d = {'K': np.array(K), 'Mk': np.array(Mk), 'L': np.array(L)}
                for k in range(K):
                    d[f'y{k}'], d[f's{k}'], d[f'n{k}'] = y[k], s[k], n[k]
                    for nm, arr in zip(names, res):
                        d[f'{nm}{k}'] = np.asarray(arr[k])
                np.savez_compressed(os.path.join(HERE, f'tango_ref_{name}.npz'), **d)
                print('tango scene', name, 'done')
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
