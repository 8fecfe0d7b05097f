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
            w = torch.empty((n, P), dtype=torch.complex64, device='cuda')          # (outputs allocated once: the launch is what is timed)
            call = lambda: eng._chk(eng.lib.disco_gevd_mwf_r1(eng.ctx, Rss.data_ptr(), Rnn.data_ptr(), n, P, 1.0, w.data_ptr(), None, None))
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            out[th] = (e0.elapsed_time(e1) / reps, w.cpu().numpy())
        d = float(abs(out[1][1] - out[0][1]).max() / abs(out[0][1]).max())
        print(f'P={P} n={n}: thread {out[1][0]:.3f} ms ({out[1][0] * 1e6 / n:.3f} ns per solve), lds group {out[0][0]:.3f} ms, max rel diff {d:.2e}', flush=True)


if __name__ == '__main__':
    main()
