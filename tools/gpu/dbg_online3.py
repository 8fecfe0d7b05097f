import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from disco_amd import _lib, synth
from disco_amd.engine import Engine
from oracle import online_oracle as oo, stft_oracle as so, mwf_oracle as mo
lib = _lib.load()
R, K, M, L, n_fft, U = 1, 1, 4, 16000, 512, 1
y, s, n = synth.make_rooms_numpy(R, K=K, M=M, L=L)
eng = Engine(lib=lib, rooms=1, nodes=1, mics=M, length=L, n_fft=n_fft)
T, F = eng.T, eng.F
mask = eng.mask_oracle(s[:, :, 0].reshape(R * K, L), n[:, :, 0].reshape(R * K, L)).reshape(R, K, T, F)
X = eng.stft(y.reshape(1, M, L)).reshape(1, 1, T, F, M)
out = eng.online_mwf(X, mask).numpy()[0, 0]                     # (T, F)
Xn, mk = X.numpy(), mask.numpy()
ref, w_all = oo.online_mwf(Xn[0, 0].transpose(2, 1, 0), mk[0, 0].T)      # (F, T)
e = np.abs(out.T - ref) / (np.abs(ref).max() + 1e-30)
print('nan in out', int(np.isnan(out.view(np.float32)).sum()), 'rel', np.linalg.norm(out.T - ref) / np.linalg.norm(ref))
bad = np.argwhere(e > 1e-3)
print('n bad', len(bad), 'frames', np.unique(bad[:, 1])[:20], 'bins', np.unique(bad[:, 0])[:40])
# eigen-gap of the offending problems, from the float64 recursion on the SAME X / mask
lam = 0.95
if len(bad):
    f0 = int(bad[0, 0]) // 16 * 16
    t0 = int(bad[0, 1])
    for f in range(f0, f0 + 16):
        Rss = np.zeros((M, M), complex); Rnn = 1e-3 * np.eye(M, dtype=complex)
        for t in range(t0 + 1):
            v = Xn[0, 0, t, f].astype(complex); m = float(mk[0, 0, t, f])
            Rss = lam * Rss + (1 - lam) * m * np.outer(v, v.conj()); Rnn = lam * Rnn + (1 - lam) * (1 - m) * np.outer(v, v.conj())
        Lc = np.linalg.cholesky(Rnn); Li = np.linalg.inv(Lc); C = Li @ Rss @ Li.conj().T
        d = np.linalg.eigvalsh(C)
        B = C / np.trace(C).real; nsq = 99
        for it in range(60):
            S2 = B @ B; tau = np.trace(S2).real; B = S2 / tau
            if 1 - tau < 1e-8:
                nsq = it + 1; break
        print(f, t0, 'ratio', d[-2] / d[-1], 'nsq', nsq, 'err', e[f, t0], 'out', out[t0, f], 'ref', ref[f, t0])
