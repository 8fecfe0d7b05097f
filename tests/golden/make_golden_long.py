#!/usr/bin/env python3
"""One LONGER, well-conditioned scene run through the reference's own `offline_tango` (same machinery as make_golden.py: the
function bodies are taken from /root/reference at run time, only librosa's stft is substituted) -> tango_ref_long.npz.

Why: the short golden scenes (17-25 frames, near-singular statistics) leave the reference's complex64 LAPACK path with
1e-4...1e-3 of its own rounding noise, so they can only be compared at 1e-2.  Here 101 frames, three microphones per node
and sensor noise at -10 dB of the sources keep every pencil well conditioned: the reference's own output is then good to
~1e-6 and the HIP path can be asserted against it DIRECTLY at the north star's 1e-4.
Runs only in the build container.  Usage: python -B tests/golden/make_golden_long.py"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as mg  # noqa: E402


def scene(rng, K, M, L):
    src_s = rng.standard_normal(L) * np.concatenate([np.zeros(L // 8), np.ones(L - L // 8)])
    src_n = rng.standard_normal(L)
    y, s, n = [], [], []
    for k in range(K):
        sk, nk = [], []
        for c in range(M):
            hs = rng.standard_normal(32) * np.exp(-np.arange(32) / 8.0)
            hn = rng.standard_normal(32) * np.exp(-np.arange(32) / 8.0)
            sk.append(0.3 * np.convolve(src_s, hs)[:L] + 0.03 * rng.standard_normal(L) * (np.arange(L) >= L // 8))
            nk.append(0.25 * np.convolve(src_n, hn)[:L] + 0.08 * rng.standard_normal(L))
        sk, nk = np.array(sk, np.float32), np.array(nk, np.float32)
        s.append(sk)
        n.append(nk)
        y.append(sk + nk)
    return y, s, n


def main():
    intern_filter, tf_mask, offline_tango, scratch = mg._load_reference()
    try:
        # The seed is the first one whose scene the reference itself computes cleanly: a single bin whose noise statistics
        # happen to be near-singular (mask ~ 1 on every frame) makes the reference's complex64 LAPACK answer for that bin
        # wrong by 1e-2 (seen with seed 20260923: cond(Rnn) = 9e7 at one bin), which says nothing about the code under test.
        # "Cleanly" = the reference's outputs agree with the float64 restatement of the same algorithm to 3e-5.
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        from oracle import tango_oracle as to
        K, M, L = 2, 3, 25600
        for seed in range(20260923, 20260923 + 50):
            rng = np.random.default_rng(seed)
            y, s, n = scene(rng, K, M, L)
            res = offline_tango(y, s, n, vads=['irm1', 'irm1'], mods=[None, None], mask_for_z='local')
            o = to.as_reference_tuple(to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], precision='f64', solver='eigh'))
            worst = max(np.linalg.norm(np.asarray(res[i][k]) - o[i][k]) / np.linalg.norm(o[i][k]) for i in range(7) for k in range(K))
            print('seed', seed, 'reference vs float64 restatement', worst)
            if worst < 3e-5:
                break
        else:
            raise SystemExit('no clean seed found')
        names = ['yf', 'sf', 'nf', 'z_y', 'z_s', 'z_n', 'zn', 'masks_z', 'mask_w']
        d = {'K': np.array(K), 'M': np.array(M), 'L': np.array(L), 'seed': np.array(seed)}
        for k in range(K):
            d[f'y{k}'], d[f's{k}'], d[f'n{k}'] = y[k], s[k], n[k]
            for nm, arr in zip(names, res):
                if nm in ('yf', 'z_y', 'zn', 'masks_z', 'sf'):
                    d[f'{nm}{k}'] = np.asarray(arr[k])
        np.savez_compressed(os.path.join(HERE, 'tango_ref_long.npz'), **d)
        print('long scene done; yf dtype', np.asarray(res[0][0]).dtype, 'shape', np.asarray(res[0][0]).shape)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
