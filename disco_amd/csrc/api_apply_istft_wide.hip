// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): the final filter + iSTFT of the wide shapes
// (P = M + K - 1 > 8) in one pass (k_apply_istft_wide, k_fused.h), reached through the whole-path entry points (api_path.hip), and of a node
// shard's step 2 on gathered z (disco_apply_istft_fused)
#include "host.h"
#include "k_fused.h"
#include "room_launch.h"

using namespace disco;
using namespace disco_host;

// shapes the one-pass kernel is built for: the wide (M, K) shapes of the room pass (the whole-path calls end in it) and the narrow 4-mic shapes
// a NODE SHARD needs it for (with all nodes of a room on the GPU those keep z on chip: k_step2_apply_istft; a shard gets z from the all-gather)
#define DISCO_FOR_WIDE_ISTFT(X_) DISCO_FOR_ROOM(X_) X_(4, 4) X_(4, 3) X_(4, 2)

namespace disco_host {
static bool wide_istft_shape(const disco_cfg& c) {
    if (c.n_fft != 512 && c.n_fft != 1024) return false;
#define X_(M_, K_) if (c.mics == M_ && c.nodes == K_) return true;
    DISCO_FOR_WIDE_ISTFT(X_)
#undef X_
    return false;
}
// does a whole-path call of this context end in the one-pass kernel?  512 / 1024-point STFT, the wide (M, K) shapes of the room pass, every node of
// a room on this GPU
bool apply_istft_wide_ok(const disco_ctx* ctx) {
    const disco_cfg& c = ctx->cfg;
    if (ctx->opt[DISCO_OPT_FUSE_WIDE_ISTFT] == 0 || sharded(ctx)) return false;
    if (c.n_fft != 512 && c.n_fft != 1024) return false;
#define X_(M_, K_) if (c.mics == M_ && c.nodes == K_) return true;
    DISCO_FOR_ROOM(X_)
#undef X_
    return false;
}

// out [R][Kl][L] = iSTFT(w^H [X; z]) (tango.py:445 + 528); yf [R][Kl][T][F] or NULL.  Honours the node shard and the z-block layout.
int apply_istft_wide(disco_ctx* ctx, const disco_c32* X, const disco_c32* Z, const disco_c32* w, disco_c32* yf, float* out, disco_stream s) {
    const disco_cfg& c = ctx->cfg;
    if (!wide_istft_shape(c)) return DISCO_E_UNSUPPORTED;
    const int K = c.nodes, WV = c.n_fft / 256;        // runs of frame pairs per workgroup (= its transform waves)
    const int n_seg = (c.length + c.hop - 1) / c.hop;
    // a workgroup = one node x WV runs of `pairs` frame pairs (2 WV filter waves + WV transform waves).  Runs as long as the signal allows while the grid keeps >= ~8 workgroups per
    // CU (a run re-reads one frame of its predecessor: 1 / (2 pairs - 1) of the traffic)
    const long long nodes = (long long)ctx->geom_rooms * ctx->Kl;
    const long long chunks_wanted = std::max<long long>(1, (8LL * ctx->n_cu + nodes - 1) / nodes);
    const int run_wanted = (int)((n_seg + chunks_wanted * WV - 1) / (chunks_wanted * WV));
    int pairs = std::max(2, (run_wanted + 2) / 2);
    if (ctx->tune_pairs > 0) pairs = ctx->tune_pairs;
    const int run_len = 2 * pairs - 1;
    const int chunks = (n_seg + WV * run_len - 1) / (WV * run_len);
    const long long items = (long long)c.rooms * ctx->Kl * chunks;
    const long long nblk = (items + 7) / 8 * 8;
    if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_enhance: batch too large for one launch");
    ApplyIstftWideArgs a;
    a.X = (const c32*)X;
    a.Z = (const c32*)Z;
    a.w = (const c32*)w;
    a.out = out;
    a.yf = (c32*)yf;
    a.T = ctx->T;
    a.L = c.length;
    a.pairs = pairs;
    a.chunks = chunks;
    a.R = c.rooms;
    a.Kl = ctx->Kl;
    a.k0 = ctx->k0;
    a.zblk = ctx->zblk;
    bool launched = false;
#define X_(M_, K_)                                                                                                                          \
    if (!launched && c.mics == M_ && K == K_) {                                                                                             \
        launched = true;                                                                                                                    \
        if (c.n_fft == 1024)                                                                                                                \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply_istft_wide<1024, M_, K_ - 1>), dim3((unsigned)nblk), dim3(768), 0, (hipStream_t)s, a,   \
                               ctx->d_win, ctx->d_tw);                                                                                      \
        else                                                                                                                                \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_apply_istft_wide<512, M_, K_ - 1>), dim3((unsigned)nblk), dim3(384), 0, (hipStream_t)s, a,    \
                               ctx->d_win, ctx->d_tw);                                                                                      \
    }
    DISCO_FOR_WIDE_ISTFT(X_)
#undef X_
    if (!launched) return DISCO_E_UNSUPPORTED;
    return check_launch(ctx, "k_apply_istft_wide");
}
}  // namespace disco_host

// disco_apply(X, Z, w, P = M + K - 1, conj_w = 1) followed by disco_istft, in one pass: the filtered spectra stay on chip
extern "C" int disco_apply_istft_fused(disco_ctx* ctx, const disco_c32* X, const disco_c32* Z, const disco_c32* w, disco_c32* yf, float* out,
                                       disco_stream s) {
    DISCO_ENTER(ctx);
    if (!X || !Z || !w || !out) return fail(ctx, DISCO_E_ARG, "disco_apply_istft_fused: null argument");
    if (!wide_istft_shape(ctx->cfg)) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_apply_istft_fused: shape not built (use disco_apply + disco_istft)");
    return STAGE(ctx, s, "apply2_istft", apply_istft_wide(ctx, X, Z, w, yf, out, s));
}
