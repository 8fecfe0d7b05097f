"""Times the cov_bench_*.so variants (built by tools/gpu/kbench/build_cov_variants.sh) on the C5 step-2 shape and checks that
they agree with each other."""
import ctypes
import glob
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
R, K, M, T, F = 200, 8, int(os.environ.get('KB_M', '8')), 313, 513
dev = torch.device('cuda:0')
torch.manual_seed(0)
X = torch.randn((R, K, T, F, M, 2), device=dev)
Z = torch.randn((R, K, T, F, 2), device=dev)
mask = torch.rand((R, K, T, F), device=dev)
NP = (M + K - 1) * (M + K) // 2
ref = None
for so in sorted(glob.glob(os.path.join(here, os.environ.get('KB_GLOB', 'cov_bench_*.so')))):
    lib = ctypes.CDLL(so)
    lib.cov_bench.restype = ctypes.c_float
    lib.cov_bench.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6
    for variant in ((0, 1) if so.endswith('_base.so') else (1,)):
        for chunks in (1, 2):
            part = torch.zeros((R * K, chunks, F, NP, 4), device=dev)
            ms = lib.cov_bench(variant, X.data_ptr(), mask.data_ptr(), Z.data_ptr(), part.data_ptr(), R, K, T, F, chunks, 5)
            torch.cuda.synchronize()
            tot = part.sum(dim=1)
            if ref is None:
                ref = tot.clone()
            # the leading 8 x 8 block is not written (SKIPLOC): compare what is
            err = float((tot - ref).abs().max() / ref.abs().max())
            print('%-28s variant %d chunks %d : %7.3f ms   max rel diff %.2e' % (os.path.basename(so), variant, chunks, ms, err), flush=True)
