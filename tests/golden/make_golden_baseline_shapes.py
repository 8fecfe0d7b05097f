#!/usr/bin/env python3
"""The reference's OWN `offline_tango` (tango.py:252-457) at BASELINE.json's shapes and full length -- one room of configs[2] (C3:
4 nodes x 4 mics, L = 160 000 samples, 626 frames) and one of configs[1] (C2: 1 node x 4 mics) -> tests/golden/tango_ref_baseline_shapes.npz.

Same machinery as make_golden.py / make_golden_scenes.py: the function bodies are taken from /root/reference at run time, only
librosa's stft is substituted.  The inputs are the bench's own synthetic rooms (disco_amd.synth.make_room_numpy: SURVEY.md 8(d) recipe,
seed 1234 + room): they are NOT stored (2 x 10 MB) but regenerated from the seed by the tests, and the fixture carries their SHA-256.
Stored: the reference's z_y and yf per node (complex64 (F, T)), and for every (node, bin) of both steps cond_2(Rnn) and the ratio d1/d0
of the two largest generalized eigenvalues (float64 restatement's matrices) -- the sensitivity data of make_golden_scenes.py, whose
KAPPA_CUT also applies here.  With 626 frames instead of 201 the statistics are far better conditioned than on the five 201-frame
scenes; this script prints the share of (node, bin)s beyond the cut and the whole-signal distance of the reference from the float64
restatement of its own algorithm.
Runs only in the build container (about 2 minutes).  Usage: python -B tests/golden/make_golden_baseline_shapes.py"""
import os
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True

SCENES = {'c3': dict(room=0, K=4, M=4), 'c2': dict(room=0, K=1, M=4)}
L_SCENE = 160000


def inputs(name):
    """The bench's synthetic room of the scene -> float32 lists [node](M, L) y, s, n (the layout offline_tango takes)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from disco_amd import synth
    sc = SCENES[name]
    y, s, n, _ = synth.make_room_numpy(sc['room'], K=sc['K'], M=sc['M'], L=L_SCENE)
    return [y[k] for k in range(sc['K'])], [s[k] for k in range(sc['K'])], [n[k] for k in range(sc['K'])]


def main():
    import make_golden as mg
    import make_golden_scenes as ms
    intern_filter, tf_mask, offline_tango, scratch = mg._load_reference()
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import tango_oracle as to
    try:
        d = {'kappa_cut': np.array(ms.KAPPA_CUT), 'L': np.array(L_SCENE)}
        for name, sc in SCENES.items():
            K, M = sc['K'], sc['M']
            y, s, n = inputs(name)
            t0 = time.time()
            res = offline_tango(y, s, n, vads=['irm1', 'irm1'], mods=[None, None], mask_for_z='local')
            t_ref = time.time() - t0
            o = to.offline_tango_vec(y, s, n, vads=['irm1', 'irm1'], precision='f64', solver='eigh')
            c1, g1, c2, g2 = ms.sensitivity(o, K)
            ok = ms.kappa(c1, g1, c2, g2) <= ms.KAPPA_CUT
            d[f'{name}_K'], d[f'{name}_M'], d[f'{name}_room'] = np.array(K), np.array(M), np.array(sc['room'])
            d[f'{name}_sha'] = np.array(ms.checksum(y, s, n))
            d[f'{name}_cond1'], d[f'{name}_cond2'] = c1.astype(np.float32), c2.astype(np.float32)
            d[f'{name}_gap1'], d[f'{name}_gap2'] = g1.astype(np.float32), g2.astype(np.float32)
            worst_in, worst_out, sig = 0.0, 0.0, 0.0
            for k in range(K):
                d[f'{name}_yf{k}'] = np.asarray(res[0][k]).astype(np.complex64)
                d[f'{name}_z_y{k}'] = np.asarray(res[3][k]).astype(np.complex64)
                for i, nm in ((0, 'yf'), (3, 'z_y')):
                    a, b = np.asarray(res[i][k]), o[nm][k]
                    e = ms.per_bin_err(a, b)
                    worst_in = max(worst_in, float(e[ok[k]].max()))
                    if (~ok[k]).any():
                        worst_out = max(worst_out, float(e[~ok[k]].max()))
                    sig = max(sig, float(np.linalg.norm(a - b) / np.linalg.norm(b)))
            print(f'{name} room {sc["room"]} K={K} M={M} T={np.asarray(res[0][0]).shape[1]}: reference ran {t_ref:.0f} s; bins beyond the cut '
                  f'{int((~ok).sum())} of {ok.size} ({100.0 * (~ok).mean():.2f} %), max cond1 {c1.max():.3g} cond2 {c2.max():.3g}; reference vs '
                  f'float64 restatement: per bin {worst_in:.2e} on the kept bins, {worst_out:.2e} on the others; WHOLE SIGNAL {sig:.2e}', flush=True)
        np.savez_compressed(os.path.join(HERE, 'tango_ref_baseline_shapes.npz'), **d)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == '__main__':
    main()
