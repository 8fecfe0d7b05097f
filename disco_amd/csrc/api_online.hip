// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): online / adaptive mode
#include "host.h"
#include "k_online.h"

using namespace disco;
using namespace disco_host;

// ---- online / adaptive mode ------------------------------------------------------------------------------

extern "C" int disco_online_mwf(disco_ctx* ctx, const disco_c32* X, const disco_c32* Z, const float* mask, int P,
                                float lambda_cor, float mu, int update_every, float init_diag, disco_c32* out,
                                disco_c32* w_last, disco_stream s) {
    DISCO_ENTER(ctx);
    const disco_cfg& c = ctx->cfg;
    if (!X || !mask || !out) return fail(ctx, DISCO_E_ARG, "disco_online_mwf: null argument");
    if (P != c.mics && P != c.mics + c.nodes - 1) return fail(ctx, DISCO_E_ARG, "disco_online_mwf: P must be mics or mics + nodes - 1");
    if (P > c.mics && !Z) return fail(ctx, DISCO_E_ARG, "disco_online_mwf: P > mics needs the exchanged z");
    if (!(lambda_cor >= 0.f && lambda_cor < 1.f) || update_every < 1 || !(init_diag > 0.f))
        return fail(ctx, DISCO_E_ARG, "disco_online_mwf: need 0 <= lambda < 1, update_every >= 1, init_diag > 0");
    if (P > 16) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_online_mwf: P must be <= 16");
    OnlineArgs a;
    a.X = (const c32*)X;
    a.Z = P > c.mics ? (const c32*)Z : nullptr;
    a.mask = mask;
    a.out = (c32*)out;
    a.w_last = (c32*)w_last;
    a.K = c.nodes;
    a.Kl = ctx->Kl;
    a.k0 = ctx->k0;
    a.T = ctx->T;
    a.F = ctx->F;
    a.M = c.mics;
    a.update_every = update_every;
    a.lambda_cor = lambda_cor;
    a.init_diag = init_diag;
    a.mu = (double)mu;
    a.n_prob = (long long)c.rooms * ctx->Kl * ctx->F;
    a.zblk = ctx->zblk;
    a.R = c.rooms;
    hipStream_t st = (hipStream_t)s;
    switch (P) {
#define C_(P_)                                                                                                          \
    case P_: {                          /* one thread per (room, node, bin) */                                          \
        constexpr int SOLVE_SMALL_THREADS = solve_small_threads<P_>();                                                  \
        const long long grid = (a.n_prob + SOLVE_SMALL_THREADS - 1) / SOLVE_SMALL_THREADS;                              \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_online_mwf_thread<P_>), dim3((unsigned)grid), dim3(SOLVE_SMALL_THREADS), 0, st, a); \
    } break;
        C_(1) C_(2) C_(3) C_(4)
        default: break;
    }
    // 5 <= P <= 7, option "solve_thread": one thread per problem as well (the float64 solve in the registers of one thread, AGPRs as its
    // second register file: k_solve_small.h)
    if (P >= 5 && P <= 7 && ctx->opt[DISCO_OPT_SOLVE_THREAD] != 0) {
        switch (P) {
            C_(5) C_(6) C_(7)
        }
        return check_launch(ctx, "k_online_mwf_thread");
    }
    if (P <= 4) return check_launch(ctx, "k_online_mwf_thread");
    switch (P) {
#undef C_
#define C_(P_)                                                                                                          \
    case P_: {                          /* a group of 8 / 16 lanes per (room, node, bin) */                             \
        const long long grid = (a.n_prob + SolveGeom<P_>::PROBS - 1) / SolveGeom<P_>::PROBS;                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_online_mwf<P_>), dim3((unsigned)grid), dim3(SolveGeom<P_>::THREADS), 0, st, a); \
    } break;
        C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
        default: return fail(ctx, DISCO_E_UNSUPPORTED, "disco_online_mwf: no kernel for this P");
    }
    return check_launch(ctx, "k_online_mwf");
}

extern "C" int disco_tango_online(disco_ctx* ctx, const float* y, const float* mask_z, const float* mask_w, float lambda_cor,
                                  int update_every, float init_diag, float* out, disco_c32* z_y, disco_c32* yf,
                                  void* workspace, size_t workspace_bytes, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!y || !mask_z || !mask_w || !out) return fail(ctx, DISCO_E_ARG, "disco_tango_online: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_online: node shard active, drive disco_online_mwf around an all-gather of z");
    const disco_cfg& c = ctx->cfg;
    const WsLayout l = ws_layout(ctx);
    char* ws = nullptr;
    int rc = acquire_ws(ctx, workspace, workspace_bytes, l, &ws, "disco_tango_online");
    if (rc) return rc;
    disco_c32* X = (disco_c32*)(ws + l.X);
    disco_c32* z = z_y ? z_y : (disco_c32*)(ws + l.z);
    disco_c32* yo = yf ? yf : (disco_c32*)(ws + l.yf);
    const int64_t G = (int64_t)c.rooms * c.nodes;
    if ((rc = STAGE(ctx, s, "stft", disco_stft(ctx, y, G, c.mics, X, s)))) return rc;
    if ((rc = STAGE(ctx, s, "online1", disco_online_mwf(ctx, X, nullptr, mask_z, c.mics, lambda_cor, c.mu, update_every, init_diag, z, nullptr, s)))) return rc;
    if (c.nodes == 1 && mask_w == mask_z) {             // nothing to append: step 2 would repeat step 1
        if (yf) HIPCHK(ctx, hipMemcpyAsync(yf, z, (size_t)G * ctx->T * ctx->F * sizeof(c32), hipMemcpyDeviceToDevice, (hipStream_t)s));
        return disco_istft(ctx, z, G, out, s);
    }
    if ((rc = STAGE(ctx, s, "online2", disco_online_mwf(ctx, X, z, mask_w, c.mics + c.nodes - 1, lambda_cor, c.mu, update_every, init_diag, yo, nullptr, s)))) return rc;
    return STAGE(ctx, s, "istft", disco_istft(ctx, yo, G, out, s));
}
