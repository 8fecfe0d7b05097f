// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): step 2 with the z exchange on chip: filter + iSTFT
#include "host.h"
#include "k_fused.h"

using namespace disco;
using namespace disco_host;

template <int M, int K>
static bool launch_apply_istft(const Step2Args& a, float* out, const float* win, const c32* tw, int L, int bpr, int pairs, dim3 grid,
                               hipStream_t st) {
    if constexpr (sizeof(ApplyIstftShared<512, M, K>) <= 160 * 1024) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_step2_apply_istft<512, M, K>), grid, dim3(64 * K), 0, st, a, out, win, tw, L, bpr, pairs);
        return true;
    } else {
        return false;
    }
}

namespace disco_host {
// does disco_step2_apply_istft_fused take this context's shape?  (512-point STFT, P <= 8, the kernel's tile within the 160 KiB LDS)
bool step2_apply_istft_ok(const disco_ctx* ctx) {
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes;
    if (c.n_fft != 512 || M + K - 1 > 8 || sharded(ctx)) return false;
#define X_(M_, KR_) if (M == M_ && K == KR_ + 1) return sizeof(ApplyIstftShared<512, M_, KR_ + 1>) <= 160 * 1024;
    DISCO_FOR_MKR(X_)
#undef X_
    return false;
}
}  // namespace disco_host

extern "C" int disco_step2_apply_istft_fused(disco_ctx* ctx, const disco_c32* X, const disco_c32* w_loc,
                                             const disco_c32* w_glo, float* out, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!X || !w_loc || !w_glo || !out) return fail(ctx, DISCO_E_ARG, "disco_step2_apply_istft_fused: null argument");
    if (sharded(ctx)) return fail(ctx, DISCO_E_UNSUPPORTED, "fused kernels need every node of a room on this GPU (node shard active)");
    const disco_cfg& c = ctx->cfg;
    const int M = c.mics, K = c.nodes, P = M + K - 1;
    if (c.n_fft != 512 || P > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_istft_fused: needs n_fft = 512 and M + K - 1 <= 8");
    Step2Args a;
    a.X = (const c32*)X;
    a.mask = nullptr;
    a.w_loc = (const c32*)w_loc;
    a.w_glo = (const c32*)w_glo;
    a.z_out = nullptr;
    a.yf = nullptr;
    a.part = nullptr;
    a.K = K;
    a.T = ctx->T;
    a.F = ctx->F;
    a.chunks = 1;
    const int n_seg = (c.length + c.hop - 1) / c.hop;
    // frame pairs per workgroup: as many as possible (<= 64) while leaving >= ~8192 waves (a workgroup has K of them)
    const long long units = (long long)ctx->geom_rooms * K;
    const long long bpr_wanted = std::max<long long>(1, (8192 + units - 1) / units);
    int pairs = (int)(((n_seg + bpr_wanted - 1) / bpr_wanted + 2) / 2);
    pairs = std::min(64, std::max(4, pairs));
    if (ctx->tune_pairs > 0) pairs = ctx->tune_pairs;
    const int bpr = (n_seg + 2 * pairs - 2) / (2 * pairs - 1);
    const long long nblk = (long long)c.rooms * bpr;
    if (nblk > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_istft_fused: batch too large");
    bool launched = false, tried = false;
#define X_(M_, KR_)                                                                                                  \
    if (!tried && M == M_ && K == KR_ + 1) {                                                                         \
        tried = true;                                                                                                \
        launched = launch_apply_istft<M_, KR_ + 1>(a, out, ctx->d_win, ctx->d_tw, c.length, bpr, pairs, dim3((unsigned)nblk), \
                                                   (hipStream_t)s);                                                  \
    }
    DISCO_FOR_MKR(X_)
#undef X_
    if (!launched) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_step2_apply_istft_fused: shape does not fit the LDS budget");
    return check_launch(ctx, "k_step2_apply_istft");
}

namespace disco_host {
// Single node, enhanced output only: iSTFT(w^H STFT(y)) straight from the samples (k_stft_apply_istft), 512-point STFT, M <= 4
int stft_apply_istft(disco_ctx* ctx, const float* y, const disco_c32* w, float* out, disco_stream s) {
    const disco_cfg& c = ctx->cfg;
    const long long G = (long long)c.rooms * c.nodes;
    const int n_seg = (c.length + c.hop - 1) / c.hop;
    const long long Gg = (long long)ctx->geom_rooms * c.nodes;
    const long long runs_wanted = std::max<long long>(1, (8192 + Gg - 1) / Gg);          // >= ~8192 waves
    int pairs = (int)(((n_seg + runs_wanted - 1) / runs_wanted + 2) / 2);
    pairs = std::min(64, std::max(4, pairs));
    if (ctx->tune_pairs > 0) pairs = ctx->tune_pairs;
    const int runs = (n_seg + 2 * pairs - 2) / (2 * pairs - 1);
    const long long items = G * runs;
    if (stft_blocks(items) > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_tango_enhance: batch too large for one launch");
    const dim3 grid((unsigned)stft_blocks(items)), block(64 * STFT_WAVES);
    switch (c.mics) {
#define C_(M_)                                                                                                          \
    case M_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_apply_istft<512, M_>), grid, block, 0, (hipStream_t)s, y, (const c32*)w, out, ctx->d_win, \
                           ctx->d_tw, c.length, ctx->T, c.pad_mode, runs, pairs, items);                                \
        break;
        C_(1) C_(2) C_(3) C_(4)
#undef C_
        default: return DISCO_E_UNSUPPORTED;
    }
    return check_launch(ctx, "k_stft_apply_istft");
}
}  // namespace disco_host
