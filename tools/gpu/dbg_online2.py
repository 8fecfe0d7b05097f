import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from disco_amd import _lib, synth
from disco_amd.engine import Engine
from oracle import online_oracle as oo, stft_oracle as so, mwf_oracle as mo
lib = _lib.load()
R, K, M, L, n_fft, U = 1, 1, 4, 16000, 512, 1
y, s, n = synth.make_rooms_numpy(R, K=K, M=M, L=L)
Y = so.stft(y[0, 0], n_fft, 256, 'reflect', np.complex128)          # M, F, T
S = so.stft(s[0, 0, 0], n_fft, 256, 'reflect', np.complex128); N = so.stft(n[0, 0, 0], n_fft, 256, 'reflect', np.complex128)
m = mo.tf_mask(S, N, 'irm1')
Tc = 48
eng = Engine(lib=lib, rooms=1, nodes=1, mics=M, length=(Tc - 1) * 256, n_fft=n_fft)
assert eng.T == Tc
X = np.ascontiguousarray(Y[:, :, :Tc].transpose(2, 1, 0)[None, None]).astype(np.complex64)     # 1,1,T,F,M
mk = np.ascontiguousarray(m[:, :Tc].T[None, None]).astype(np.float32)
out, w = eng.online_mwf(X, mk, want_w=True)
out, w = out.numpy()[0, 0], w.numpy()[0, 0]                  # (T,F), (F,P)
ref_out, w_all = oo.online_mwf(X[0, 0].transpose(2, 1, 0), mk[0, 0].T)
w_ref = w_all[:, Tc - 1]
for f in range(156, 180):
    e = np.linalg.norm(w[f] - w_ref[f]) / np.linalg.norm(w_ref[f])
    print(f, 'w relerr %.2e' % e, 'gpu', np.round(w[f][:2], 4), 'ref', np.round(w_ref[f][:2], 4), 'ratio |w|', np.linalg.norm(w[f]) / np.linalg.norm(w_ref[f]))
