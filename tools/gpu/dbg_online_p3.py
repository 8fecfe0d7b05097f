import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from disco_amd import _lib
from disco_amd.engine import Engine
lib = _lib.load()
g = np.load('tests/golden/online_ref.npz')
for tag in ('p3', 'p5u4'):
    V, mask, ref, w_ref = g[tag + '_V'], g[tag + '_mask'], g[tag + '_out'], g[tag + '_w']
    lam, mu, init, U = (float(x) for x in g[tag + '_params'])
    P, F, T = V.shape
    eng = Engine(lib=lib, rooms=1, nodes=1, mics=P, length=(T - 1) * 256, n_fft=512)
    X = np.zeros((1, 1, T, eng.F, P), np.complex64)
    X[0, 0, :, :F] = V.transpose(2, 1, 0)
    mk = np.full((1, 1, T, eng.F), 0.5, np.float32)
    mk[0, 0, :, :F] = mask.T
    out, w = eng.online_mwf(X, mk, lambda_cor=lam, mu=mu, update_every=int(U), init_diag=init, want_w=True)
    out, w = out.numpy()[0, 0, :, :F].T, w.numpy()[0, 0, :F]
    print(tag, 'P', P, 'F', F, 'T', T, 'lam mu init U', lam, mu, init, U)
    e = np.abs(out - ref)
    print(' per-frame max err (first 12):', e.max(axis=0)[:12], ' ref scale', np.abs(ref).max())
    print(' per-bin rel err:', np.linalg.norm(out - ref, axis=1) / np.linalg.norm(ref, axis=1))
    print(' w[0]:', w[0], ' w_ref[0,-1]:', w_ref[0, -1])
    # the same pencils through the batch solver: smoothed matrices of the last frame from the oracle recursion
    Rs = np.zeros((F, P, P), complex); Rn = np.tile(np.eye(P) * init, (F, 1, 1)).astype(complex)
    for t in range(T):
        v = V[:, :, t].T                      # (F, P)
        vv = v[:, :, None] * v[:, None, :].conj()
        Rs = lam * Rs + (1 - lam) * mask[:, t][:, None, None] * vv
        Rn = lam * Rn + (1 - lam) * (1 - mask[:, t])[:, None, None] * vv
    eng2 = Engine(lib=lib, rooms=1, nodes=1, mics=P, length=5120, n_fft=512)
    wb, _ = eng2.gevd_mwf_r1(Rs.astype(np.complex64)[None, None][:, :, :, :, :].repeat(1, 0).reshape(1, 1, F, P, P)[:, :, :eng2.F] if F == eng2.F else np.concatenate([Rs.astype(np.complex64), np.tile(np.eye(P, dtype=np.complex64), (eng2.F - F, 1, 1))])[None, None],
                             (Rn.astype(np.complex64) if F == eng2.F else np.concatenate([Rn.astype(np.complex64), np.tile(np.eye(P, dtype=np.complex64), (eng2.F - F, 1, 1))]))[None, None], mu=mu)
    print(' batch solver on the last frame\'s pencils: w[0]', wb.numpy()[0, 0, 0])
