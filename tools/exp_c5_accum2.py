"""Which float32 sums of the C5 covariances cost the accuracy, and what summation order buys it back?  (CPU experiment, test
infrastructure: uses the oracle.)  The float64 oracle of the 2-iteration scheme is run with its covariances formed the way a
GPU lane forms them -- `chunks` time chunks per node, inside a chunk sequential float32 accumulation of blocks of `block`
frames that were themselves summed sequentially from zero, chunks combined in float64, rounded to complex64 -- per block class:
    s1     the M x M statistics of step 1 (also the leading block of every step-2 pencil: SKIPLOC)
    cross  the y-z entries of step 2,   zz  the z-z entries of step 2
Usage: python tools/exp_c5_accum2.py <room> s1=<c>:<b>[u]|x|d cross=... zz=...      (x = exact sums rounded to complex64; r = float64 sums of the complex64-ROUNDED rows; d = exact sums in
float64, unrounded; a trailing u = the float32 partial sums combined in float64 and handed over unrounded)
Results: profiles/r04_c5_accumulation.txt"""
import os
import sys

import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco_amd import synth
from oracle import tango_oracle as to
K, M, N, L = 8, 8, 1024, 160000
room = int(sys.argv[1])
spec = dict(a.split('=') for a in sys.argv[2:])
y, s, n, _ = synth.make_room_numpy(room, K=K, M=M, L=L)
s2 = np.zeros_like(y); n2 = np.zeros_like(y); s2[:, 0] = s[:, 0]; n2[:, 0] = n[:, 0]
def run():
    o = to.offline_tango_vec(y, s2, n2, vads=['irm1', 'irm1'], n_fft=N, hop=N // 2, precision='f64', solver='eigh', extra_iters=1)
    return [np.asarray(o['yf'][k]) for k in range(K)]
ref = run()
orig = to._cov_mean

def emul(Vt, chunks, block, rounded=True):
    F, T, P = Vt.shape
    V32 = Vt.astype(np.complex64)
    tot = np.zeros((F, P, P), np.complex128)
    for c in range(chunks):
        t0, t1 = T * c // chunks, T * (c + 1) // chunks
        acc = np.zeros((F, P, P), np.complex64)
        for tb in range(t0, t1, block):
            blk = np.zeros((F, P, P), np.complex64)
            for t in range(tb, min(tb + block, t1)):
                blk += V32[:, t, :, None] * np.conjugate(V32[:, t, None, :])
            acc += blk
        tot += acc
    return (tot / T).astype(np.complex64).astype(np.complex128) if rounded else tot / T

def emul_sub(Vt, sub, rounded=False, f64_tree=False):
    """the room pass's order (k_room.h): `sub` interleaved float32 accumulators (frame t goes to accumulator t % sub, each sums its frames
    sequentially), combined pairwise in float32 (the lane-level halvings), ONE block: handed to the solver as it is."""
    F, T, P = Vt.shape
    V32 = Vt.astype(np.complex64)
    acc = [np.zeros((F, P, P), np.complex64) for _ in range(sub)]
    for t in range(T):
        acc[t % sub] += V32[:, t, :, None] * np.conjugate(V32[:, t, None, :])
    if f64_tree:
        acc = [sum(a.astype(np.complex128) for a in acc)]
    while len(acc) > 1:
        h = len(acc) // 2
        acc = [acc[i] + acc[i + h] for i in range(h)]
    tot = acc[0].astype(np.complex128) / T
    return tot.astype(np.complex64).astype(np.complex128) if rounded else tot


PERM = int(spec.pop('perm', 0))       # > 0: the frames are summed in a random order (seed): another realisation of the same roundings' statistics
ROT = int(spec.pop('rot', 0))         # 1: the node's own M rows in the basis of the DFT across its (circular) array
def cov(V, ref32):
    Vt = np.transpose(V, (1, 2, 0))
    F, T, P = Vt.shape
    if PERM:
        Vt = Vt[:, np.random.default_rng(PERM).permutation(T)]
    U = np.eye(P, dtype=np.complex128)
    if ROT:
        U[:M, :M] = np.fft.fft(np.eye(M)) / np.sqrt(M)
        Vt = Vt @ U.T                                      # rows -> U v
    Vx = np.transpose(Vt, (2, 0, 1))
    exact64 = orig(Vx, False)
    exact = exact64.astype(np.complex64).astype(np.complex128)
    R = exact.copy()
    cache = {}
    def get(sp):
        if sp == 'x':
            return exact
        if sp == 'd':                      # round 5: float64 sums handed to the solver UNROUNDED ((hi, lo) at the solver's door)
            return exact64
        if sp == 'r':                      # inputs rounded to complex64 (the spectra and z the GPU holds), products and sums in float64, unrounded
            if 'r' not in cache:
                V64 = Vt.astype(np.complex64).astype(np.complex128)
                cache['r'] = np.einsum('ftp,ftq->fpq', V64, np.conjugate(V64)) / T
            return cache['r']
        if sp.startswith('s'):             # "sN": N interleaved accumulators, float32 tree, one block (the room pass)
            if sp not in cache:                # "sNu": the N accumulators combined in FLOAT64 and handed over unrounded ((hi, lo) blocks)
                cache[sp] = emul_sub(Vt, int(sp[1:].rstrip('u')), f64_tree=sp.endswith('u'))
            return cache[sp]
        if sp not in cache:
            unrounded = sp.endswith('u')   # "c:bu": float32 partial sums combined in float64 and NOT rounded to complex64
            c, b = (int(v) for v in sp.rstrip('u').split(':'))
            cache[sp] = emul(Vt, c, b, rounded=not unrounded)
        return cache[sp]
    R[:, :M, :M] = get(spec['s1'])[:, :M, :M]
    if P > M:
        R[:, :M, M:] = get(spec['cross'])[:, :M, M:]
        R[:, M:, :M] = get(spec['cross'])[:, M:, :M]
        R[:, M:, M:] = get(spec['zz'])[:, M:, M:]
    if ROT:
        R = np.conjugate(U.T) @ R @ U                      # back, in float64 (the solver would simply work in the rotated basis)
    return R
to._cov_mean = cov
got = run()
err = [float('%.2e' % (np.linalg.norm(got[k] - ref[k]) / np.linalg.norm(ref[k]))) for k in range(K)]
print(room, ' '.join(sys.argv[2:]), 'max %.2e' % max(err), err, flush=True)
