"""Where does the C5 error of room 199 come from?  (CPU experiment, test infrastructure: uses the oracle.)
The float64 oracle of the 2-iteration scheme is run with ONE ingredient degraded at a time and compared with itself:
  x32           the STFTs rounded to complex64 (what the HIP path stores), everything else float64
  seq32         the covariances accumulated sequentially in float32 over the frames (what a GPU lane does)
  seq32_diag64  the same, diagonal entries exact;   seq32_off64  the same, off-diagonal entries exact
  round32       exact covariances rounded to complex64 (what the solver is handed)
Usage: python tools/exp_c5_accumulation.py <mode>      (about 3 minutes per mode; results: profiles/r03_c5_accumulation.txt)"""
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_amd import synth
from oracle import stft_oracle as so, tango_oracle as to
K, M, N, L = 8, 8, 1024, 160000
y, s, n, _ = synth.make_room_numpy(199, K=K, M=M, L=L)
s2 = np.zeros_like(y); n2 = np.zeros_like(y); s2[:, 0] = s[:, 0]; n2[:, 0] = n[:, 0]
def run():
    o = to.offline_tango_vec(y, s2, n2, vads=['irm1', 'irm1'], n_fft=N, hop=N // 2, precision='f64', solver='eigh', extra_iters=1)
    return [np.asarray(o['yf'][k]) for k in range(K)]
ref = run()
orig = to._cov_mean
mode = sys.argv[1]
if mode == 'x32':
    stft0 = so.stft
    so.stft = to.so.stft = lambda x, *a, **kw: (lambda X: X.astype(np.complex64).astype(X.dtype))(stft0(x, *a, **kw))
def cov32(V, ref32):
    Vt = np.transpose(V, (1, 2, 0))
    F, T, P = Vt.shape
    if mode == 'seq32':            # sequential float32 accumulation over the frames (what a GPU lane does), inputs rounded to c64
        V32 = Vt.astype(np.complex64)
        acc = np.zeros((F, P, P), np.complex64)
        for t in range(T):
            acc += V32[:, t, :, None] * np.conjugate(V32[:, t, None, :])
        return (acc / np.float32(T)).astype(np.complex128)
    if mode == 'seq32_diag64':     # float32 sequential accumulation off the diagonal, exact diagonal
        V32 = Vt.astype(np.complex64)
        acc = np.zeros((F, P, P), np.complex64)
        for t in range(T):
            acc += V32[:, t, :, None] * np.conjugate(V32[:, t, None, :])
        R = (acc / np.float32(T)).astype(np.complex128)
        ex = orig(V, False)
        idx = np.arange(P)
        R[:, idx, idx] = ex[:, idx, idx].astype(np.complex64).astype(np.complex128)
        return R
    if mode == 'seq32_off64':      # exact off-diagonal, float32 sequential diagonal
        V32 = Vt.astype(np.complex64)
        acc = np.zeros((F, P, P), np.complex64)
        for t in range(T):
            acc += V32[:, t, :, None] * np.conjugate(V32[:, t, None, :])
        R32 = (acc / np.float32(T)).astype(np.complex128)
        R = orig(V, False).astype(np.complex64).astype(np.complex128)
        idx = np.arange(P)
        R[:, idx, idx] = R32[:, idx, idx]
        return R
    if mode == 'round32':          # exact sums, result rounded to complex64 (what the solver is handed)
        return orig(V, False).astype(np.complex64).astype(np.complex128)
if mode != 'x32':
    to._cov_mean = cov32
got = run()
print(mode, [float('%.2e' % (np.linalg.norm(got[k] - ref[k]) / np.linalg.norm(ref[k]))) for k in range(K)], flush=True)
