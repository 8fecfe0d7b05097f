import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from disco_amd import _lib, synth
from disco_amd.engine import Engine
from oracle import online_oracle as oo
lib = _lib.load()
for (R, K, M, L, n_fft, U) in [(2, 4, 4, 12000, 512, 1), (1, 1, 4, 16000, 512, 1), (1, 5, 1, 8000, 512, 1)]:
    y, s, n = synth.make_rooms_numpy(R, K=K, M=M, L=L)
    eng = Engine(lib=lib, rooms=R, nodes=K, mics=M, length=L, n_fft=n_fft)
    T, F = eng.T, eng.F
    mask = eng.mask_oracle(s[:, :, 0].reshape(R * K, L), n[:, :, 0].reshape(R * K, L)).reshape(R, K, T, F).numpy()
    runs = []
    for i in range(4):
        out, z, yf = eng.tango_online(y, mask, update_every=U)
        runs.append((z.numpy(), yf.numpy()))
    print('config', (R, K, M, L, n_fft, U), 'T', T)
    for i in range(1, 4):
        for nm, a, b in (('z', runs[0][0], runs[i][0]), ('yf', runs[0][1], runs[i][1])):
            d = np.argwhere(a != b)
            if len(d):
                print('  run', i, nm, 'differs at', len(d), 'elements; rooms', np.unique(d[:, 0]), 'nodes', np.unique(d[:, 1]),
                      'frames min', d[:, 2].min(), 'bins', np.unique(d[:, 3])[:40], 'n bins', len(np.unique(d[:, 3])))
    for r in range(R):
        o = oo.online_tango(y[r], s[r], n[r], n_fft=n_fft, hop=n_fft // 2, update_every=U)
        for k in range(K):
            for nm, got, ref in (('z', runs[0][0][r, k].T, o['z'][k]), ('yf', runs[0][1][r, k].T, o['yf'][k])):
                rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
                if rel > 1e-4:
                    e = np.abs(got - ref) / (np.abs(ref).max() + 1e-30)
                    bad = np.argwhere(e > 1e-3)
                    print('  BAD room', r, 'node', k, nm, 'rel', rel, 'n bad', len(bad), 'bins', np.unique(bad[:, 0])[:30], 'first frame', bad[:, 1].min() if len(bad) else None)
