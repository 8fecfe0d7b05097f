// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): the final filter + iSTFT of the wide shapes
// (P = M + K - 1 > 8) in one pass (k_apply_istft_wide, k_fused.h), reached through the whole-path entry points (api_path.hip)
#include "host.h"
#include "k_fused.h"
#include "room_launch.h"

using namespace disco;
using namespace disco_host;

namespace disco_host {
// does the one-pass kernel take this context's shape?  512 / 1024-point STFT, the wide (M, K) shapes of the room pass, every node of a room on
// this GPU
bool apply_istft_wide_ok(const disco_ctx* ctx) {
    const disco_cfg& c = ctx->cfg;
    if (ctx->opt[DISCO_OPT_FUSE_WIDE_ISTFT] == 0 || sharded(ctx)) return false;
    if (c.n_fft != 512 && c.n_fft != 1024) return false;
#define X_(M_, K_) if (c.mics == M_ && c.nodes == K_) return true;
    DISCO_FOR_ROOM(X_)
#undef X_
    return false;
}

// out [R][K][L] = iSTFT(w^H [X; z]) (tango.py:445 + 528); yf [R][K][T][F] or NULL
int apply_istft_wide(disco_ctx* ctx, const disco_c32* X, const disco_c32* Z, const disco_c32* w, disco_c32* yf, float* out, disco_stream s) {
    const disco_cfg& c = ctx->cfg;
    if (!apply_istft_wide_ok(ctx)) return DISCO_E_UNSUPPORTED;
    const int K = c.nodes, WV = c.n_fft / 256;        // runs of frame pairs per workgroup (= its transform waves)
    const int n_seg = (c.length + c.hop - 1) / c.hop;
    // a workgroup = one node x WV runs of `pairs` frame pairs (2 WV filter waves + WV transform waves).  Runs as long as the signal allows while the grid keeps >= ~8 workgroups per
    // CU (a run re-reads one frame of its predecessor: 1 / (2 pairs - 1) of the traffic)
    const long long nodes = (long long)ctx->geom_rooms * K;
    const long long chunks_wanted = std::max<long long>(1, (8LL * ctx->n_cu + nodes - 1) / nodes);
    const int run_wanted = (int)((n_seg + chunks_wanted * WV - 1) / (chunks_wanted * WV));
    int pairs = std::max(2, (run_wanted + 2) / 2);
    if (ctx->tune_pairs > 0) pairs = ctx->tune_pairs;
    const int run_len = 2 * pairs - 1;
    const int chunks = (n_seg + WV * run_len - 1) / (WV * run_len);
    const long long items = (long long)c.rooms * K * chunks;
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
    DISCO_FOR_ROOM(X_)
#undef X_
    if (!launched) return DISCO_E_UNSUPPORTED;
    return check_launch(ctx, "k_apply_istft_wide");
}
}  // namespace disco_host
