// Debug harness: k_online_mwf_thread<P> / k_online_mwf<P> outside the library (variants via -D flags).
#include <hip/hip_runtime.h>
#include "../../../disco_amd/csrc/k_online.h"
using namespace disco;
#ifndef KB_P
#define KB_P 3
#endif
extern "C" int online_dbg(const void* X, const void* mask, void* out, void* w_last, int T, int F, float lam, float init, double mu, int U) {
    OnlineArgs a{};
    a.X = (const c32*)X; a.Z = nullptr; a.mask = (const float*)mask; a.out = (c32*)out; a.w_last = (c32*)w_last;
    a.K = 1; a.Kl = 1; a.k0 = 0; a.T = T; a.F = F; a.M = KB_P; a.update_every = U; a.lambda_cor = lam; a.init_diag = init; a.mu = mu;
    a.n_prob = F; a.zblk = 1; a.R = 1;
    const unsigned nblk = (unsigned)((F + SOLVE_SMALL_THREADS - 1) / SOLVE_SMALL_THREADS);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_online_mwf_thread<KB_P>), dim3(nblk), dim3(SOLVE_SMALL_THREADS), 0, 0, a);
    hipDeviceSynchronize();
    return (int)hipGetLastError();
}
