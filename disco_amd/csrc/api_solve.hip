// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): rank-1 GEVD-MWF solves
#include "host.h"
#include "k_solve.h"
#include "k_solve_small.h"

using namespace disco;
using namespace disco_host;

namespace disco_host {
// api_solve_dpp.hip: 9 <= P <= 16 in registers (k_solve_dpp.h)
void launch_solve_dpp(int P, const SolveSrc& src, long long n_prob, double mu, c32* w, c32* t1, hipStream_t s);
}

template <int P>
static void launch_solve(const SolveSrc& src, long long n_prob, double mu, c32* w, c32* t1, hipStream_t s, bool thread) {
    if (P <= 4 || (P <= 8 && thread)) {             // one thread per pencil (k_solve_small.h)
        if constexpr (P <= 8) {
        constexpr int SOLVE_SMALL_THREADS = solve_small_threads<P>();
        const long long grid = (n_prob + SOLVE_SMALL_THREADS - 1) / SOLVE_SMALL_THREADS;
        if (src.part)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1_thread<P, true>), dim3((unsigned)grid), dim3(SOLVE_SMALL_THREADS), 0, s, src, n_prob,
                               mu, w, t1);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1_thread<P, false>), dim3((unsigned)grid), dim3(SOLVE_SMALL_THREADS), 0, s, src, n_prob,
                               mu, w, t1);
        }
        return;
    }
    if constexpr (P >= 5) {
    const int probs = SolveGeom<P>::PROBS;
    const long long grid = (n_prob + probs - 1) / probs;
    if (src.part)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1<P, true>), dim3((unsigned)grid), dim3(SolveGeom<P>::THREADS), 0, s, src,
                           n_prob, mu, w, t1);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gevd_mwf_r1<P, false>), dim3((unsigned)grid), dim3(SolveGeom<P>::THREADS), 0, s, src,
                           n_prob, mu, w, t1);
    }
}

static int solve_dispatch(disco_ctx* ctx, const SolveSrc& src, int64_t n_prob, int P, float mu, disco_c32* w, disco_c32* t1,
                          disco_stream s) {
    if (P < 1 || P > 16) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_gevd_mwf_r1: P must be in 1..16");
    if (n_prob / 4 > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_gevd_mwf_r1: batch too large");
    hipStream_t st = (hipStream_t)s;
    if (P >= 9 && ctx->opt[DISCO_OPT_SOLVE_DPP] != 0) {
        launch_solve_dpp(P, src, n_prob, (double)mu, (c32*)w, (c32*)t1, st);
        return check_launch(ctx, "k_gevd_mwf_r1_dpp");
    }
    switch (P) {
#define C_(P_) case P_: launch_solve<P_>(src, n_prob, (double)mu, (c32*)w, (c32*)t1, st, ctx->opt[DISCO_OPT_SOLVE_THREAD] != 0); break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
    }
    return check_launch(ctx, "k_gevd_mwf_r1");
}

extern "C" int disco_gevd_mwf_r1(disco_ctx* ctx, const disco_c32* Rss, const disco_c32* Rnn, int64_t n_prob, int P,
                                 float mu, disco_c32* w, disco_c32* t1, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!Rss || !Rnn || !w || n_prob < 1) return fail(ctx, DISCO_E_ARG, "disco_gevd_mwf_r1: bad argument");
    SolveSrc src;
    src.Rss = (const c32*)Rss;
    src.Rnn = (const c32*)Rnn;
    src.part = nullptr;
    src.F = 1;
    src.chunks = 1;
    src.inv_T = 1.f;
    src.part_loc = nullptr;
    src.chunks_loc = 0;
    src.M_loc = 0;
    return solve_dispatch(ctx, src, n_prob, P, mu, w, t1, s);
}
extern "C" int disco_gevd_mwf_r1_pending(disco_ctx* ctx, float mu, disco_c32* w, disco_c32* t1, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!w) return fail(ctx, DISCO_E_ARG, "disco_gevd_mwf_r1_pending: null argument");
    if (ctx->pending_chunks < 1 || !ctx->scratch)
        return fail(ctx, DISCO_E_ARG, "disco_gevd_mwf_r1_pending: no covariance call has left partial sums in this context");
    SolveSrc src;
    src.Rss = nullptr;
    src.Rnn = nullptr;
    src.part = (const float4*)(ctx->pending_skiploc ? ctx->scratch2 : ctx->scratch);
    src.F = ctx->F;
    src.chunks = ctx->pending_chunks;
    // The partial sums go to the solvers UNSCALED (round 5): w and t1 do not change when Rxx and Rnn are scaled together, and without the
    // multiplication by 1 / T an entry that arrives as one float32 block (the fused step-2 pass leaves one block per node) or as the (hi, lo)
    // pair of a float64 total (k_cov_loc_f64, the room pass) reaches the float64 arithmetic exactly as it was summed; a product with 1 / T
    // cost every entry a rounding (k_solve_dpp.h)
    src.inv_T = 1.0f;
    src.part_loc = ctx->pending_skiploc ? (const float4*)ctx->scratch : nullptr;
    src.chunks_loc = ctx->pending_skiploc ? ctx->loc_chunks : 0;
    src.M_loc = ctx->pending_skiploc ? ctx->loc_M : 0;
    return solve_dispatch(ctx, src, (int64_t)ctx->cfg.rooms * ctx->Kl * ctx->F, ctx->pending_P, mu, w, t1, s);
}
