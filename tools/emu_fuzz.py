#!/usr/bin/env python3
"""Random-shape fuzzing of the kernel sources on the hipemu CPU build (test tooling; no GPU, nothing here is a product path).

    python tools/emu_fuzz.py SEED SECONDS

Draws (rooms, nodes, mics, length, n_fft, fused/staged) at random and runs the parity checks of tests/parity_checks.py
against the float64 oracle: the whole path, the fused step-2 kernels, the covariance/solve/apply stages with every mask_for_z
data flow, the 9 <= P <= 16 kernels, the iterated scheme, the online recursion, the RIR convolution and the image-source generator.  Prints one line per case; FAIL lines carry the error dict.
Known benign failures: the `mask_max` tail bound at bins where |N| ~ 0, and solver agreement between two accumulation orders
when a case has barely more frames than channels (ill-conditioned covariances).  Round 1: 500+ cases, no kernel defect."""
import os
import sys
import time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import emu_build, parity_checks as pc
from disco_amd import synth
from disco_amd.engine import Engine
lib = emu_build.load_emu()
mk = lambda **c: Engine(lib=lib, **c)
rng = np.random.default_rng(int(sys.argv[1]))
t_end = time.time() + float(sys.argv[2])
while time.time() < t_end:
    kind = rng.choice(['path', 'big', 'step2', 'reuse', 'csa', 'iter', 'online', 'conv', 'ism'])
    try:
        if kind == 'path':     # whole path, P <= 9
            K = int(rng.integers(1, 6)); M = int(rng.integers(1, 6)); n_fft = int(rng.choice([512, 1024])); R = int(rng.integers(1, 4))
            L = (M + K + 2) * n_fft // 2 + int(rng.integers(0, 6000)); staged = bool(rng.integers(0, 2)); tag = (R, K, M, L, n_fft, staged)
            y, s, n = synth.make_rooms_numpy(R, K=K, M=M, L=L)
            r = pc.check_tango_end_to_end(mk, y, s, n, n_fft=n_fft, tol=1e-4, staged_step2=staged)
        elif kind == 'big':    # 9 <= P <= 16, staged kernels
            while True:
                K = int(rng.integers(2, 7)); M = int(rng.integers(2, 9)); P = M + K - 1
                if 9 <= P <= 16: break
            L = (P + 4) * 256 + int(rng.integers(0, 3000))
            y, s, n = synth.make_rooms_numpy(1, K=K, M=M, L=L)
            r = pc.check_tango_end_to_end(mk, y, s, n, n_fft=512, tol=1e-4, staged_step2=bool(rng.integers(0, 2)))
            tag = (K, M, L)
        elif kind == 'step2':
            while True:
                K = int(rng.integers(1, 9)); M = int(rng.integers(1, 8))
                if M + K - 1 <= 8: break
            R = int(rng.integers(1, 4)); L = 2560 + int(rng.integers(0, 3000)); tag = (R, K, M, L)
            r = pc.check_step2_fused(mk, R=R, K=K, M=M, L=L)
        elif kind == 'reuse':
            while True:
                K = int(rng.integers(2, 9)); M = int(rng.integers(1, 8))
                if M + K - 1 <= 8: break
            R = int(rng.integers(1, 4)); L = 2560 + int(rng.integers(0, 3000)); tag = (R, K, M, L)
            r = pc.check_step2_reuse(mk, R=R, K=K, M=M, L=L)
        elif kind == 'csa':
            K = int(rng.integers(1, 6)); M = int(rng.integers(1, 6)); R = int(rng.integers(1, 4))
            sz = bool(rng.integers(0, 2)); mr = bool(rng.integers(0, 2)); L = 3072 + int(rng.integers(0, 2000)); tag = (R, K, M, L, sz, mr)
            r = pc.check_cov_solve_apply(mk, R=R, K=K, M=M, L=L, same_z=sz, mask_remote=mr)
        elif kind == 'online':
            K = int(rng.integers(1, 4)); M = int(rng.integers(1, 4)); U = int(rng.integers(1, 5)); n_fft = int(rng.choice([512, 1024]))
            L = n_fft * int(rng.integers(2, 5)) + int(rng.integers(0, n_fft // 4)); tag = (K, M, L, n_fft, U)
            r = pc.check_online_mwf(mk, R=1, K=K, M=M, L=L, n_fft=n_fft, update_every=U)
        elif kind == 'conv':
            Ld = int(rng.integers(8, 5000)); Lh = int(rng.integers(6, 3000)); n_ch = int(rng.integers(1, 5)); n_sig = int(rng.integers(1, 3))
            out_len = None if rng.integers(0, 2) else int(rng.integers(Ld // 7 + 8, Ld + Lh + 500)); tag = (n_sig, n_ch, Ld, Lh, out_len)
            r = pc.check_rir_convolve(mk, n_sig=n_sig, n_ch=n_ch, Ld=Ld, Lh=Lh, out_len=out_len)
        elif kind == 'ism':
            mo = int(rng.integers(0, 5)); rl = int(rng.choice([2048, 4096])); tag = (mo, rl)
            r = pc.check_ism_rir(mk, n_room=int(rng.integers(1, 3)), S=int(rng.integers(1, 3)), Q=int(rng.integers(1, 4)), max_order=mo,
                                 rir_len=rl, seed=int(rng.integers(0, 1000)))
        else:
            K = int(rng.integers(2, 5)); M = int(rng.integers(1, 4)); it = int(rng.integers(2, 4)); L = 4096 + int(rng.integers(0, 3000)); tag = (K, M, L, it)
            r = pc.check_iterated_outputs(mk, K, M, L, 512, it)
        print('ok', kind, tag, flush=True)
    except Exception as ex:
        print('FAIL', kind, tag, repr(ex)[:300], flush=True)
