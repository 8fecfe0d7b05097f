// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): STFT + step-1 covariance in one pass
#include "host.h"
#include "k_stft.h"

using namespace disco;
using namespace disco_host;

// ---------------------------------------------------------------------------------------------------------
// STFT + step-1 covariance in one pass
// ---------------------------------------------------------------------------------------------------------
template <int N, bool STORE = true>
static bool launch_stft_cov(int M, dim3 grid, hipStream_t st, const float* y, const float* mask, c32* X, float4* part,
                            const float* win, const c32* tw, int L, int T, int pad_mode, int chunks, int runw) {
    const dim3 block(64 * STFT_WAVES);
    switch (M) {
#define C_(M_)                                                                                                          \
    case M_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_cov<N, M_, STORE>), grid, block, 0, st, y, mask, X, part, win, tw, L, T, pad_mode, \
                           chunks, runw);                                                                               \
        return true;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6)
#undef C_
    }
    if constexpr (N == 512) {          // the 1024-point spectrum tile of 7-8 mics does not fit the 160 KiB LDS
        switch (M) {
#define C_(M_)                                                                                                          \
    case M_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_cov<N, M_, STORE>), grid, block, 0, st, y, mask, X, part, win, tw, L, T, pad_mode, \
                           chunks, runw);                                                                               \
        return true;
            C_(7) C_(8)
#undef C_
        }
    }
    return false;
}

// frame chunks (= workgroups per node) of the fused STFT + covariance pass and the frames each of its waves streams: runs as long
// as possible while leaving >= ~2048 workgroups for the chip, but NOT LONGER THAN DISCO_STFT_COV_RUN frames: a workgroup's fold sums the
// frames of its four waves' runs in float32, and the length of that sum is what the distance of the C3 output from the float64 oracle follows
// (the whole-batch sweep's worst room of 1000, output against the oracle: 4.1e-5 at 80 frames per wave, 2.3e-5 at 40, 1.3e-5 at 20 -- every room halves
// with the run, profiles/r05_t_runw_parity.txt), while the pass itself is as fast at 40 or 20 as at 80 (profiles/r05_t_runw_speed.txt); what
// more blocks cost is the solvers' fetch (k_solve_small.h: four blocks of an entry in flight together).
#ifndef DISCO_STFT_COV_RUN
#define DISCO_STFT_COV_RUN 40
#endif
namespace disco_host {
int stft_cov_chunks(const disco_ctx* ctx, int* runw_out) {
    const long long G = (long long)ctx->geom_rooms * ctx->cfg.nodes;
    const long long chunks_wanted = std::max<long long>(1, (2048 + G - 1) / G);
    int runw = (int)((ctx->T + STFT_WAVES * chunks_wanted - 1) / (STFT_WAVES * chunks_wanted));
    runw = std::min(DISCO_STFT_COV_RUN, std::max(8, runw));
    if (ctx->tune_runw > 0) runw = ctx->tune_runw;
    if (runw_out) *runw_out = runw;
    return (ctx->T + STFT_WAVES * runw - 1) / (STFT_WAVES * runw);
}

// store = false (internal, single-node path): the spectra are not written (X may be NULL); only for shapes the fused kernel takes
int stft_cov_partials(disco_ctx* ctx, const float* y, const float* mask_z, disco_c32* X, int* chunks_out, disco_stream s, bool store) {
    if (!y || !mask_z || (store && !X)) return fail(ctx, DISCO_E_ARG, "disco_stft_cov_fused: null argument");
    // (works on a node shard too: nothing in this pass looks beyond one node -- X, masks and partial sums then hold the shard's Kl nodes per room)
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics;
    if (M > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft_cov_fused: more than 8 mics per node");
    if (c.n_fft == 1024 && M > 6) {        // staged form of the same two operations
        // (Round 3 built the one-pass form for this shape -- four transform waves with a channel pair each plus eight fold waves, one
        // bin per thread, two LDS tiles, one barrier per frame: parity-green and SLOWER, 12.8 ms against 5.1 + 3.9 ms per C5 launch.
        // The 144 accumulator registers per bin force 12 waves and 100 KiB of LDS into one workgroup, i.e. ONE workgroup and four
        // transform waves per CU where k_stft_pairs keeps eight; the transforms set the pace.  Dropped, see DESIGN.md.)
        if (!store) return fail(ctx, DISCO_E_UNSUPPORTED, "stft_cov without store: shape needs the staged kernels");
        int rc0 = STAGE(ctx, s, "stft", disco_stft(ctx, y, (int64_t)c.rooms * ctx->Kl, M, X, s));
        if (rc0) return rc0;
        return STAGE(ctx, s, "cov1", cov_partials(ctx, X, mask_z, nullptr, nullptr, 0, M, chunks_out, s));
    }
    const long long G = (long long)c.rooms * ctx->Kl;
    int runw = 0;
    const int chunks = stft_cov_chunks(ctx, &runw);
    const int NP = M * (M + 1) / 2;
    int rc = ensure_scratch(ctx, (size_t)G * chunks * ctx->F * NP * sizeof(float4));
    if (rc) return rc;
    if (G * chunks > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft_cov_fused: batch too large");
    const dim3 grid((unsigned)(G * chunks));
    const bool ok = !store
        ? STAGE(ctx, s, "stft_cov1_nostore", c.n_fft == 512
            ? (launch_stft_cov<512, false>(M, grid, (hipStream_t)s, y, mask_z, nullptr, (float4*)ctx->scratch, ctx->d_win, ctx->d_tw, c.length,
                                           ctx->T, c.pad_mode, chunks, runw))
            : (launch_stft_cov<1024, false>(M, grid, (hipStream_t)s, y, mask_z, nullptr, (float4*)ctx->scratch, ctx->d_win, ctx->d_tw, c.length,
                                            ctx->T, c.pad_mode, chunks, runw)))
        : STAGE(ctx, s, "stft_cov1", c.n_fft == 512
        ? launch_stft_cov<512>(M, grid, (hipStream_t)s, y, mask_z, (c32*)X, (float4*)ctx->scratch, ctx->d_win, ctx->d_tw, c.length,
                               ctx->T, c.pad_mode, chunks, runw)
        : launch_stft_cov<1024>(M, grid, (hipStream_t)s, y, mask_z, (c32*)X, (float4*)ctx->scratch, ctx->d_win, ctx->d_tw, c.length,
                                ctx->T, c.pad_mode, chunks, runw));
    if (!ok) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft_cov_fused: unsupported mic count");
    *chunks_out = chunks;
    ctx->pending_chunks = chunks;
    ctx->pending_P = M;
    ctx->pending_skiploc = 0;
    ctx->loc_chunks = chunks;          // kept for a possible re-use by step 2 of the same disco_tango_enhance call
    ctx->loc_M = M;
    ctx->loc_X = X;
    ctx->loc_mask = mask_z;
    if (!store || sharded(ctx)) ctx->loc_M = 0;       // nothing to pair these partial sums with later (a shard runs the staged step 2)
    return check_launch(ctx, "k_stft_cov");
}

}  // namespace disco_host

extern "C" int disco_stft_cov_fused(disco_ctx* ctx, const float* y, const float* mask_z, disco_c32* X, disco_c32* Rss,
                                    disco_c32* Rnn, disco_stream s) {
    DISCO_ENTER(ctx);
    if ((Rss == nullptr) != (Rnn == nullptr)) return fail(ctx, DISCO_E_ARG, "disco_stft_cov_fused: Rss and Rnn must both be given or both be NULL");
    int chunks = 1;
    int rc = stft_cov_partials(ctx, y, mask_z, X, &chunks, s);
    if (rc || !Rss) return rc;
    return cov_finalize(ctx, chunks, ctx->cfg.mics, Rss, Rnn, s);
}
