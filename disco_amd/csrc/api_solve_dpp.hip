// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): launches of the register / DPP form of the
// rank-1 GEVD-MWF solve for 9 <= P <= 16 (k_solve_dpp.h).  A unit of its own: eight sizes x two sources of fully unrolled code.
#include "host.h"
#include "k_solve_dpp.h"

using namespace disco;

namespace disco_host {

template <int P>
static void launch(const SolveSrc& src, long long n_prob, double mu, c32* w, c32* t1, hipStream_t s) {
    const long long grid = (n_prob + DppSolveGeom<P>::PROBS - 1) / DppSolveGeom<P>::PROBS;
    if (src.part)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1_dpp<P, true>), dim3((unsigned)grid), dim3(DppSolveGeom<P>::THREADS), 0, s, src, n_prob, mu,
                           w, t1);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1_dpp<P, false>), dim3((unsigned)grid), dim3(DppSolveGeom<P>::THREADS), 0, s, src, n_prob, mu,
                           w, t1);
}

void launch_solve_dpp(int P, const SolveSrc& src, long long n_prob, double mu, c32* w, c32* t1, hipStream_t s) {
    switch (P) {
#define C_(P_) case P_: launch<P_>(src, n_prob, mu, w, t1, s); break;
        C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
    }
}

}  // namespace disco_host

extern "C" int disco_selftest_dpp(disco_ctx* ctx, const double* a, const double* b, int64_t n, double* out_hw, double* out_ref, disco_stream s) {
    DISCO_ENTER(ctx);
    static_assert(DPP_SELFTEST_OPS == DISCO_DPP_SELFTEST_OPS, "header and kernel agree");
    if (!a || !b || !out_hw || !out_ref || n < 64 || n % 64) return fail(ctx, DISCO_E_ARG, "disco_selftest_dpp: bad argument (n: a multiple of 64)");
    hipLaunchKernelGGL(k_dpp_selftest, dim3((unsigned)(n / 64)), dim3(64), 0, (hipStream_t)s, (const c64*)a, (const c64*)b, (c64*)out_hw, (c64*)out_ref);
    return check_launch(ctx, "k_dpp_selftest");
}
