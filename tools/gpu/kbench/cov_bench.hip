// Micro-benchmark of the P = 15 covariance kernels (C5 shape) outside the library: build variants with -D flags, time with
// HIP events.  tools/gpu/kbench/run_cov_bench.py drives it.
#include <hip/hip_runtime.h>
#include "../../../disco_amd/csrc/k_cov.h"
using namespace disco;

#ifndef KB_M
#define KB_M 8
#endif
#ifndef KB_KR
#define KB_KR 7
#endif

extern "C" float cov_bench(int variant, const void* X, const void* mask, const void* Z, void* part, int R, int K, int T, int F,
                           int chunks, int reps) {
    CovArgs a{};
    a.X = (const c32*)X;
    a.mask = (const float*)mask;
    a.Zs = a.Zn = (const c32*)Z;
    a.part = (float4*)part;
    a.K = K; a.T = T; a.F = F; a.chunks = chunks; a.mask_remote = 1; a.Kl = K; a.k0 = 0; a.zblk = K; a.R = R;
    const int tiles = (F - 1 + 63) / 64;
    unsigned nblk = (unsigned)((long long)R * K * (tiles + 1) * chunks);
    const unsigned nblk_lds = DISCO_COV_XCD ? (nblk + 7) / 8 * 8 : nblk;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto launch = [&]() {
        if (variant == 0)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_split<KB_M, KB_KR, true>), dim3(nblk), dim3(64 * cov_split_waves<KB_KR, true>()), 0, 0, a);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_cov_split_lds<KB_M, KB_KR, true>), dim3(nblk_lds), dim3(64 * cov_split_waves<KB_KR, true>()), 0, 0, a);
    };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? ms / reps : -1.f;
}
