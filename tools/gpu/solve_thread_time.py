#!/usr/bin/env python3
"""Per-launch time of the rank-1 GEVD-MWF solve on full matrices for 5 <= P <= 8: the LDS group solver against one THREAD per pencil
(option "solve_thread").  Usage: solve_thread_time.py [n_prob] [P ...]"""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit('/tools/', 1)[0])
from disco_amd.engine import Engine
from tools.gpu.solve_time import pencils


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1028000
    sizes = [int(x) for x in sys.argv[2:]] or [5, 6, 7, 8]
    for P in sizes:
        Rss, Rnn = pencils(n, P)
        out = {}
        for th in (1, 0):
            eng = Engine(rooms=1, nodes=1, mics=1, length=1024)
            eng.set_option('solve_thread', th)
            for _ in range(2):
                w, _t = eng.gevd_mwf_r1(Rss, Rnn, want_t1=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                w, _t = eng.gevd_mwf_r1(Rss, Rnn, want_t1=False)
            torch.cuda.synchronize()
            out[th] = ((time.perf_counter() - t0) / reps * 1e3, w.numpy())
        d = float(abs(out[1][1] - out[0][1]).max() / abs(out[0][1]).max())
        print(f'P={P} n={n}: thread {out[1][0]:.3f} ms ({out[1][0] * 1e6 / n:.3f} ns per solve), lds group {out[0][0]:.3f} ms, max rel diff {d:.2e}', flush=True)


if __name__ == '__main__':
    main()
