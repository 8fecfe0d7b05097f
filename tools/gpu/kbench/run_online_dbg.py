import ctypes, glob, os, sys
import numpy as np
import torch
here = os.path.dirname(os.path.abspath(__file__))
g = np.load(os.path.join(here, '..', '..', '..', 'tests', 'golden', 'online_ref.npz'))
tag = 'p3'
V, mask, ref, w_ref = g[tag + '_V'], g[tag + '_mask'], g[tag + '_out'], g[tag + '_w']
lam, mu, init, U = (float(x) for x in g[tag + '_params'])
P, F, T = V.shape
X = torch.from_numpy(np.ascontiguousarray(V.transpose(2, 1, 0)).astype(np.complex64)).cuda()     # [T][F][P]
mk = torch.from_numpy(np.ascontiguousarray(mask.T).astype(np.float32)).cuda()                    # [T][F]
for so in sorted(glob.glob(os.path.join(here, 'online_dbg_*.so'))):
    lib = ctypes.CDLL(so)
    lib.online_dbg.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_double, ctypes.c_int]
    out = torch.zeros((T, F), dtype=torch.complex64, device='cuda')
    w = torch.zeros((F, P), dtype=torch.complex64, device='cuda')
    rc = lib.online_dbg(X.data_ptr(), mk.data_ptr(), out.data_ptr(), w.data_ptr(), T, F, lam, init, mu, int(U))
    o = out.cpu().numpy().T
    print(os.path.basename(so), 'rc', rc, 'rel err out', np.linalg.norm(o - ref) / np.linalg.norm(ref), 'w[0]', w.cpu().numpy()[0], flush=True)
print('w_ref[0]', w_ref[0, -1])
