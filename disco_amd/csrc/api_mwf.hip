// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): intern_filter's other two branches
#include "host.h"
#include "k_solve.h"

using namespace disco;
using namespace disco_host;

// The two branches of intern_filter the hot path never takes (see k_mwf_variants)
extern "C" int disco_mwf_filter(disco_ctx* ctx, const disco_c32* Rxx, const disco_c32* Rnn, int64_t n_prob, int P, float mu,
                                int type, disco_c32* w, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!Rxx || !Rnn || !w || n_prob < 1) return fail(ctx, DISCO_E_ARG, "disco_mwf_filter: bad argument");
    if (type != DISCO_FILTER_R1_MWF && type != DISCO_FILTER_MWF) return fail(ctx, DISCO_E_ARG, "disco_mwf_filter: unknown filter type");
    if (P < 1 || P > 16) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_mwf_filter: P must be in 1..16");
    if (n_prob / 4 > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_mwf_filter: batch too large");
    hipStream_t st = (hipStream_t)s;
    switch (P) {
#define C_(P_)                                                                                                          \
    case P_: {                                                                                                          \
        const dim3 grid((unsigned)((n_prob + SolveGeom<P_>::PROBS - 1) / SolveGeom<P_>::PROBS)), block(SolveGeom<P_>::THREADS);  \
        if (type == DISCO_FILTER_MWF)                                                                                   \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mwf_variants<P_, FILTER_MWF>), grid, block, 0, st, (const c32*)Rxx, (const c32*)Rnn, \
                               (long long)n_prob, (double)mu, (c32*)w);                                                 \
        else                                                                                                            \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mwf_variants<P_, FILTER_R1_MWF>), grid, block, 0, st, (const c32*)Rxx, (const c32*)Rnn, \
                               (long long)n_prob, (double)mu, (c32*)w);                                                 \
    } break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
    }
    return check_launch(ctx, "k_mwf_variants");
}
