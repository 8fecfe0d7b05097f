// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): filter-and-sum
#include "host.h"
#include "k_apply.h"
#include "k_cov.h"

using namespace disco;
using namespace disco_host;

extern "C" int disco_apply(disco_ctx* ctx, const disco_c32* X, const disco_c32* Z, const disco_c32* w, int P, int conj_w,
                           disco_c32* out, disco_stream s) {
    DISCO_ENTER(ctx);
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, KR = P - M;
    if (!X || !w || !out) return fail(ctx, DISCO_E_ARG, "disco_apply: null argument");
    if (KR != 0 && KR != c.nodes - 1) return fail(ctx, DISCO_E_ARG, "disco_apply: P must be M or M + K - 1");
    if (KR > 0 && !Z) return fail(ctx, DISCO_E_ARG, "disco_apply: Z required when P > M");
    if (P > CB_PMAX) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_apply: P > 16 not supported");
    const long long G = (long long)c.rooms * ctx->Kl;
    const long long TF = (long long)ctx->T * ctx->F;
    int bpn = (int)std::min<long long>((TF + 255) / 256, 64);
    while ((long long)bpn * G > 0x7fffffffLL && bpn > 1) bpn >>= 1;
    const dim3 grid((unsigned)(G * bpn)), block(256);
    bool launched = false;
#define X_(M_, KR_)                                                                                                  \
    if (!launched && M == M_ && KR == KR_) {                                                                         \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply<M_, KR_>), grid, block, 0, (hipStream_t)s, (const c32*)X,         \
                           (const c32*)Z, (const c32*)w, (c32*)out, c.nodes, ctx->T, ctx->F, conj_w, bpn, ctx->Kl,   \
                           ctx->k0, ctx->zblk, (long long)c.rooms);                                                                                 \
        launched = true;                                                                                             \
    }
    DISCO_FOR_MKR(X_)
#undef X_
    if (!launched) {
        const int tiles = (ctx->F + 63) / 64;
        const long long Gg = (long long)ctx->geom_rooms * ctx->Kl;
#ifndef DISCO_APPLY_ITEMS
#define DISCO_APPLY_ITEMS 131072            // workgroups aimed at: more, shorter frame runs keep the nodes of a room in step (their remote rows meet in L2): 3 -> 10 chunks at C5 read 21.7 instead of 25.5 GB (profiles/r03_r_*)
#endif
        int t_chunks = (int)std::min<long long>(std::max<long long>(1, (DISCO_APPLY_ITEMS + Gg * tiles - 1) / (Gg * tiles)), std::max(1, ctx->T / 8));
        while (G * tiles * t_chunks > 0x7ffffff0LL && t_chunks > 1) t_chunks >>= 1;
        const long long items_m = G * tiles * t_chunks;
        const dim3 grid_m((unsigned)((items_m + DISCO_APPLY_XCD - 1) / DISCO_APPLY_XCD * DISCO_APPLY_XCD));      // ids are dealt over the XCDs
        if ((M == 4 || M == 8) && KR >= 1) {        // contiguous granule loads through a wave-private LDS tile (k_apply_mq)
            const int krt = KR <= 1 ? 1 : (KR <= 3 ? 3 : (KR <= 7 ? 7 : 15));
#define Q_(M_, KRT_)                                                                                                                  \
    if (M == M_ && krt == KRT_)                                                                                                       \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply_mq<M_, KRT_>), grid_m, dim3(64), 0, (hipStream_t)s, (const c32*)X, (const c32*)Z,  \
                           (const c32*)w, (c32*)out, KR, c.nodes, ctx->T, ctx->F, conj_w, tiles, t_chunks, ctx->Kl, ctx->k0, ctx->zblk, \
                           (long long)c.rooms);
            Q_(4, 1) Q_(4, 3) Q_(4, 7) Q_(4, 15) Q_(8, 1) Q_(8, 3) Q_(8, 7) Q_(8, 15)
#undef Q_
            return check_launch(ctx, "k_apply_mq");
        }
        switch (M) {
#define C_(M_)                                                                                                          \
    case M_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply_m<M_>), grid_m, dim3(64), 0, (hipStream_t)s, (const c32*)X, (const c32*)Z, \
                           (const c32*)w, (c32*)out, KR, c.nodes, ctx->T, ctx->F, conj_w, tiles, t_chunks, ctx->Kl, ctx->k0, ctx->zblk, \
                           (long long)c.rooms);                                                                    \
        break;
            C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8)
#undef C_
            default: return fail(ctx, DISCO_E_UNSUPPORTED, "disco_apply: more than 8 mics per node");
        }
    }
    return check_launch(ctx, "k_apply");
}

extern "C" int disco_noise_residual(disco_ctx* ctx, const disco_c32* X, const disco_c32* z, disco_c32* zn, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!X || !z || !zn) return fail(ctx, DISCO_E_ARG, "disco_noise_residual: null argument");
    const disco_cfg& c = ctx->cfg;
    const long long n = (long long)c.rooms * ctx->Kl * ctx->T * ctx->F;
    hipLaunchKernelGGL(k_noise_residual, dim3((unsigned)std::min<long long>((n + 255) / 256, 16384)), dim3(256), 0,
                       (hipStream_t)s, (const c32*)X, (const c32*)z, (c32*)zn, n, c.mics, c.ref_mic);
    return check_launch(ctx, "k_noise_residual");
}
// Local part of a P-entry filter (the iterated scheme's re-compression filter); honours the node shard, so the node-sharded
// driver can run the DANSE-style iterations with one all-gather of z per iteration.
extern "C" int disco_filter_head(disco_ctx* ctx, const disco_c32* w_glo, int P, disco_c32* w_loc, disco_stream s) {
    DISCO_ENTER(ctx);
    const disco_cfg& c = ctx->cfg;
    if (!w_glo || !w_loc) return fail(ctx, DISCO_E_ARG, "disco_filter_head: null argument");
    if (P < c.mics) return fail(ctx, DISCO_E_ARG, "disco_filter_head: P < M");
    const long long nb = (long long)c.rooms * ctx->Kl * ctx->F;
    hipLaunchKernelGGL(k_filter_head, dim3((unsigned)std::min<long long>((nb * c.mics + 255) / 256, 65535)), dim3(256), 0, (hipStream_t)s,
                       (const c32*)w_glo, (c32*)w_loc, nb, c.mics, P);
    return check_launch(ctx, "k_filter_head");
}
