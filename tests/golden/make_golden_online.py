#!/usr/bin/env python3
"""Golden fixture for the online (exponentially smoothed) mode, produced by RUNNING THE REFERENCE'S OWN PRIMITIVES.

The reference ships the smoothing update `spatial_correlation_matrix` (se_utils/internal_formulas.py:84-103) and the
filter `intern_filter(..., 'gevd', rank=1)` (:56-73) but no loop around them.  This
script drives the two *imported reference functions* frame by frame in the obvious recursion

    Rss <- spatial_correlation_matrix(Rss, v_t, lambda, M = m_t)          (mask on the mixture, as its docstring says)
    Rnn <- spatial_correlation_matrix(Rnn, v_t, lambda, M = 1 - m_t)
    every `update_every` frames:  w <- intern_filter(Rss, Rnn, mu, 'gevd', 1)[0]
    out_t = w^H v_t

and stores inputs and outputs, so that oracle/online_oracle.py (and through it the HIP kernel) is pinned on the
reference's arithmetic for both primitives.  Runs only in the build container.   python -B tests/golden/make_golden_online.py
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True
REF = '/root/reference'


def main():
    scratch = tempfile.mkdtemp(prefix='disco_ref_')
    shutil.copytree(os.path.join(REF, 'disco_theque'), os.path.join(scratch, 'disco_theque'))
    sys.path.insert(0, scratch)
    from disco_theque.se_utils.internal_formulas import intern_filter, spatial_correlation_matrix   # reference code

    rng = np.random.default_rng(77)
    out = {}
    for tag, P, F, T, U in (('p3', 3, 5, 40, 1), ('p5u4', 5, 3, 48, 4)):
        # a rank-1 "speech" component plus diffuse noise, speech active on a random subset of frames
        a = rng.standard_normal((F, P)) + 1j * rng.standard_normal((F, P))
        src = (rng.standard_normal((F, T)) + 1j * rng.standard_normal((F, T))) * (rng.random((1, T)) > 0.3)
        noise = 0.4 * (rng.standard_normal((P, F, T)) + 1j * rng.standard_normal((P, F, T)))
        V = (a.T[:, :, None] * src[None] + noise).astype(np.complex64)
        S2 = np.abs(a.T[0][:, None] * src) ** 2
        N2 = np.abs(noise[0]) ** 2
        mask = (S2 / (S2 + N2 + 1e-12)).astype(np.float32)
        lam, mu, init = 0.95, 1.0, 1e-3
        y = np.zeros((F, T), np.complex128)
        w_all = np.zeros((F, T, P), np.complex128)
        for f in range(F):
            Rss = np.zeros((P, P), np.complex128)
            Rnn = init * np.eye(P, dtype=np.complex128)
            w = np.zeros(P, np.complex128)
            for t in range(T):
                v = V[:, f, t].astype(np.complex128)
                Rss = spatial_correlation_matrix(Rss, v, lam, M=float(mask[f, t]))
                Rnn = spatial_correlation_matrix(Rnn, v, lam, M=1.0 - float(mask[f, t]))
                if t % U == 0:
                    w = intern_filter(Rss, Rnn, mu=mu, type='gevd', rank=1)[0]
                w_all[f, t] = w
                y[f, t] = np.conjugate(w) @ v
        out.update({f'{tag}_V': V, f'{tag}_mask': mask, f'{tag}_out': y, f'{tag}_w': w_all,
                    f'{tag}_params': np.array([lam, mu, init, U])})
    np.savez_compressed(os.path.join(HERE, 'online_ref.npz'), **out)
    shutil.rmtree(scratch, ignore_errors=True)
    print('wrote online_ref.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
