// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): step 2 with the z exchange on chip: covariances
#include "host.h"
#include "k_fused.h"

using namespace disco;
using namespace disco_host;

// ---------------------------------------------------------------------------------------------------------
// step 2 with the in-register z exchange
// ---------------------------------------------------------------------------------------------------------
namespace disco_host {
int step2_chunks(const disco_ctx* ctx, int tiles_plus_1) {
    const long long base = (long long)ctx->geom_rooms * tiles_plus_1;
    long long c = (4096 + base - 1) / base;
    if (c > 8) c = 8;
    if (ctx->tune_step2_chunks > 0) c = ctx->tune_step2_chunks;
    if (c > ctx->T) c = ctx->T;
    if (c < 1) c = 1;
    return (int)c;
}

// skiploc: the caller guarantees that `scratch` still holds the step-1 partial sums of THIS X with THIS mask (only
// disco_tango_enhance can know); the leading M x M block is then neither accumulated nor written and the step-2 partials
// go to `scratch2`.
int step2_cov_partials(disco_ctx* ctx, const disco_c32* X, const float* mask_w, const disco_c32* w_loc,
                       disco_c32* z_out, int* chunks_out, disco_stream s, bool skiploc) {
    if (!X || !mask_w || !w_loc) return fail(ctx, DISCO_E_ARG, "disco_step2_cov_fused: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "fused kernels need every node of a room on this GPU (node shard active)");
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes, P = M + K - 1;
    if (P > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_cov_fused: M + K - 1 > 8 not supported yet");
    const int tiles = (ctx->F - 1) / 64;
    const int chunks = step2_chunks(ctx, tiles + 1);
    const long long G = (long long)c.rooms * K;
    const int NP = P * (P + 1) / 2;
    const size_t need = (size_t)G * chunks * ctx->F * NP * sizeof(float4);
    int rc = 0;
    rc = skiploc ? ensure_scratch2(ctx, need) : ensure_scratch(ctx, need);
    if (rc) return rc;
    Step2Args a;
    a.X = (const c32*)X;
    a.mask = mask_w;
    a.w_loc = (const c32*)w_loc;
    a.w_glo = nullptr;
    a.z_out = (c32*)z_out;
    a.yf = nullptr;
    a.part = (float4*)(skiploc ? ctx->scratch2 : ctx->scratch);
    a.K = K;
    a.T = ctx->T;
    a.F = ctx->F;
    a.chunks = chunks;
    const long long nblk = (long long)c.rooms * (tiles + 1) * chunks;
    if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_cov_fused: batch too large");
    bool launched = false;
#define X_(M_, KR_)                                                                                                  \
    if (!launched && M == M_ && K == KR_ + 1) {                                                                      \
        if (skiploc)                                                                                                 \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step2_cov_fused<M_, KR_ + 1, true>), dim3((unsigned)nblk),          \
                               dim3(64 * (KR_ + 1)), 0, (hipStream_t)s, a);                                          \
        else                                                                                                         \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step2_cov_fused<M_, KR_ + 1, false>), dim3((unsigned)nblk),         \
                               dim3(64 * (KR_ + 1)), 0, (hipStream_t)s, a);                                          \
        launched = true;                                                                                             \
    }
    DISCO_FOR_MKR(X_)
#undef X_
    if (!launched) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_cov_fused: unsupported (M, K) combination");
    *chunks_out = chunks;
    ctx->pending_chunks = chunks;
    ctx->pending_P = P;
    ctx->pending_skiploc = skiploc ? 1 : 0;
    if (!skiploc) ctx->loc_M = 0;
    return check_launch(ctx, "k_step2_cov_fused");
}

}  // namespace disco_host

extern "C" int disco_step2_cov_fused_reuse(disco_ctx* ctx, const disco_c32* X, const float* mask_w, const disco_c32* w_loc,
                                           disco_c32* z_out, disco_stream s) {
    DISCO_ENTER(ctx);
    if (ctx->loc_M != ctx->cfg.mics || ctx->cfg.nodes < 2 || ctx->Kl != ctx->cfg.nodes)
        return fail(ctx, DISCO_E_ARG, "disco_step2_cov_fused_reuse: no step-1 partial sums of disco_stft_cov_fused are held by this context");
    if (ctx->loc_X != X || ctx->loc_mask != mask_w)
        return fail(ctx, DISCO_E_ARG, "disco_step2_cov_fused_reuse: X / mask_w are not the arrays the held step-1 partial sums were computed from");
    int chunks = 1;
    return step2_cov_partials(ctx, X, mask_w, w_loc, z_out, &chunks, s, true);
}
extern "C" int disco_step2_cov_fused(disco_ctx* ctx, const disco_c32* X, const float* mask_w, const disco_c32* w_loc,
                                     disco_c32* z_out, disco_c32* Rss, disco_c32* Rnn, disco_stream s) {
    DISCO_ENTER(ctx);
    if ((Rss == nullptr) != (Rnn == nullptr)) return fail(ctx, DISCO_E_ARG, "disco_step2_cov_fused: Rss and Rnn must both be given or both be NULL");
    int chunks = 1;
    int rc = step2_cov_partials(ctx, X, mask_w, w_loc, z_out, &chunks, s);
    if (rc || !Rss) return rc;
    return cov_finalize(ctx, chunks, ctx->cfg.mics + ctx->cfg.nodes - 1, Rss, Rnn, s);
}
