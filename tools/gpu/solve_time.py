#!/usr/bin/env python3
"""Per-launch time of the rank-1 GEVD-MWF solve on full matrices (disco_gevd_mwf_r1: no partial-sum loads), LDS group solver against
the register / DPP form (option "solve_dpp"), on covariance-like pencils.  Usage: solve_time.py [n_prob] [P ...]"""
import sys
import time

import torch

from disco_amd.engine import Engine


def pencils(n, P, T=64, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    def cn(*s):
        return torch.complex(torch.randn(*s, generator=g, device='cuda'), torch.randn(*s, generator=g, device='cuda'))
    a = cn(n, P, 1)
    X = a * cn(n, 1, T) + 0.3 * cn(n, P, T)
    N = cn(n, P, T)
    return (X @ X.conj().transpose(1, 2) / T).to(torch.complex64).contiguous(), (N @ N.conj().transpose(1, 2) / T).to(torch.complex64).contiguous()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 820800
    sizes = [int(x) for x in sys.argv[2:]] or [15]
    for P in sizes:
        Rss, Rnn = pencils(n, P)
        out = {}
        for dpp in (1, 0):
            eng = Engine(rooms=1, nodes=1, mics=1, length=1024)
            eng.set_option('solve_dpp', dpp)
            for _ in range(2):
                w, _t = eng.gevd_mwf_r1(Rss, Rnn, want_t1=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                w, _t = eng.gevd_mwf_r1(Rss, Rnn, want_t1=False)
            torch.cuda.synchronize()
            out[dpp] = ((time.perf_counter() - t0) / reps * 1e3, w.numpy())
        d = float(abs(out[1][1] - out[0][1]).max() / abs(out[0][1]).max())
        print(f'P={P} n={n}: dpp {out[1][0]:.3f} ms, lds {out[0][0]:.3f} ms, max rel diff {d}', flush=True)


if __name__ == '__main__':
    main()
