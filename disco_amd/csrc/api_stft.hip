// libdisco_hip.so -- host side of the C ABI declared in include/disco_hip.h (gfx950 only): STFT / iSTFT / masks
#include "host.h"
#include "k_stft.h"

using namespace disco;
using namespace disco_host;

// ---------------------------------------------------------------------------------------------------------
// STFT family
// ---------------------------------------------------------------------------------------------------------

template <int N>
static bool launch_stft(int chp, dim3 grid, hipStream_t st, const float* x, c32* X, const float* win, const c32* tw, int chans,
                        int L, int T, int pad_mode, int runs, long long n_items) {
    const dim3 block(64 * STFT_WAVES);
    // one channel pair per wave (k_stft_pairs) where k_stft's all-pairs-in-registers form drops to one wave per SIMD (measured:
    // N = 1024 from 2 pairs on, N = 512 from 3); the grid is (group, run) then, not waves
    if (chp >= (N == 1024 ? 2 : 3) && chp <= STFT_WAVES && n_items <= 0x7fffffffLL) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft_pairs<N>), dim3((unsigned)n_items), block, 0, st, x, X, win, tw, chans, L, T, pad_mode,
                           runs);
        return true;
    }
    switch (chp) {
#define C_(P_)                                                                                                          \
    case P_:                                                                                                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stft<N, P_>), grid, block, 0, st, x, X, win, tw, chans, L, T, pad_mode, runs, n_items); \
        return true;
        C_(1) C_(2)
#undef C_
    }
    return false;
}

// the transform of signals of ANY length L (T = 1 + L / hop frames) with this context's window, FFT size and padding: disco_stft passes the
// cfg's length, the streaming online path the length of a chunk's transform block (api_online.hip)
namespace disco_host {
int stft_any(disco_ctx* ctx, const float* x, int64_t n_sig, int chans, disco_c32* X, int L, int T, disco_stream s) {
    if (!x || !X || n_sig < 1 || chans < 1) return fail(ctx, DISCO_E_ARG, "disco_stft: bad argument");
    if (chans > 8) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft: more than 8 channels per signal group");
    const int runs = stft_runs(T);
    const long long n_items = (long long)n_sig * runs;
    if (stft_blocks(n_items) > 0x7fffffffLL)
        return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft: batch too large for one launch");
    const disco_cfg& c = ctx->cfg;
    const dim3 grid((unsigned)stft_blocks(n_items));
    const int chp = (chans + 1) / 2;
    const bool ok = c.n_fft == 512
        ? launch_stft<512>(chp, grid, (hipStream_t)s, x, (c32*)X, ctx->d_win, ctx->d_tw, chans, L, T, c.pad_mode, runs, n_items)
        : launch_stft<1024>(chp, grid, (hipStream_t)s, x, (c32*)X, ctx->d_win, ctx->d_tw, chans, L, T, c.pad_mode, runs, n_items);
    if (!ok) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_stft: unsupported channel count");
    return check_launch(ctx, "k_stft");
}
}  // namespace disco_host

extern "C" int disco_stft(disco_ctx* ctx, const float* x, int64_t n_sig, int chans, disco_c32* X, disco_stream s) {
    DISCO_ENTER(ctx);
    return stft_any(ctx, x, n_sig, chans, X, ctx->cfg.length, ctx->T, s);
}

extern "C" int disco_mask_oracle(disco_ctx* ctx, const float* s_ref, const float* n_ref, int64_t n_sig, float* mask,
                                 disco_stream s) {
    DISCO_ENTER(ctx);
    if (!s_ref || !n_ref || !mask || n_sig < 1) return fail(ctx, DISCO_E_ARG, "disco_mask_oracle: bad argument");
    const disco_cfg& c = ctx->cfg;
    if (c.mask_type < DISCO_MASK_IRM || c.mask_type > DISCO_MASK_IAM)
        return fail(ctx, DISCO_E_ARG, "disco_mask_oracle: unknown mask type");
    const int runs = stft_runs(ctx->T);
    const long long n_items = (long long)n_sig * runs;
    if (stft_blocks(n_items) > 0x7fffffffLL)
        return fail(ctx, DISCO_E_UNSUPPORTED, "disco_mask_oracle: batch too large for one launch");
    const float thr = powf(10.f, c.mask_bin_thr_db / 10.f);                 // math_utils.py db2lin (power)
    const dim3 grid((unsigned)stft_blocks(n_items));
    StageScope stage_scope_(ctx, s, "mask_oracle");
    if (c.n_fft == 512)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mask_oracle<512>), grid, dim3(64 * STFT_WAVES), 0,
                           (hipStream_t)s, s_ref, n_ref, mask, ctx->d_win, ctx->d_tw, c.length, ctx->T, c.pad_mode,
                           c.mask_type, c.mask_pow, thr, runs, n_items);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_mask_oracle<1024>), grid, dim3(64 * STFT_WAVES), 0,
                           (hipStream_t)s, s_ref, n_ref, mask, ctx->d_win, ctx->d_tw, c.length, ctx->T, c.pad_mode,
                           c.mask_type, c.mask_pow, thr, runs, n_items);
    return check_launch(ctx, "k_mask_oracle");
}

extern "C" int disco_tf_mask(disco_ctx* ctx, const disco_c32* S, const disco_c32* N, int64_t n_elem, int mask_type,
                             int mask_pow, float bin_thr_db, float* mask, disco_stream s) {
    DISCO_ENTER(ctx);
    if (!S || !N || !mask || n_elem < 1 || mask_pow < 0) return fail(ctx, DISCO_E_ARG, "disco_tf_mask: bad argument");
    if (mask_type < DISCO_MASK_IRM || mask_type > DISCO_MASK_IAM) return fail(ctx, DISCO_E_ARG, "disco_tf_mask: unknown mask type");
    const unsigned grid = (unsigned)std::min<long long>((n_elem + 255) / 256, 8192);
    hipLaunchKernelGGL(k_tf_mask, dim3(grid), dim3(256), 0, (hipStream_t)s, (const c32*)S, (const c32*)N, mask,
                       (long long)n_elem, mask_type, mask_pow, powf(10.f, bin_thr_db / 10.f));
    return check_launch(ctx, "k_tf_mask");
}

namespace disco_host {
// solo: one frame per inverse transform (k_stft.h: what the online entry points use, so that a stream of chunks equals the whole clip bit for bit)
int istft_any(disco_ctx* ctx, const disco_c32* Z, int64_t n_sig, float* out, int L, int T, disco_stream s, bool solo) {
    if (!Z || !out || n_sig < 1) return fail(ctx, DISCO_E_ARG, "disco_istft: bad argument");
    const disco_cfg& c = ctx->cfg;
    const int n_seg = (L + c.hop - 1) / c.hop;
    const int segs = solo ? STFT_WAVES - 1 : ISTFT_SEGS;
    const int bps = (n_seg + segs - 1) / segs;
    const long long grid = (long long)n_sig * bps;
    if (grid > 0x7fffffffLL) return fail(ctx, DISCO_E_UNSUPPORTED, "disco_istft: batch too large for one launch");
    const dim3 gr((unsigned)grid), bl(64 * STFT_WAVES);
    hipStream_t st = (hipStream_t)s;
    if (c.n_fft == 512 && !solo) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_istft<512, false>), gr, bl, 0, st, (const c32*)Z, out, ctx->d_win, ctx->d_tw, L, T, bps);
    else if (c.n_fft == 512) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_istft<512, true>), gr, bl, 0, st, (const c32*)Z, out, ctx->d_win, ctx->d_tw, L, T, bps);
    else if (!solo) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_istft<1024, false>), gr, bl, 0, st, (const c32*)Z, out, ctx->d_win, ctx->d_tw, L, T, bps);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_istft<1024, true>), gr, bl, 0, st, (const c32*)Z, out, ctx->d_win, ctx->d_tw, L, T, bps);
    return check_launch(ctx, "k_istft");
}
}  // namespace disco_host

extern "C" int disco_istft(disco_ctx* ctx, const disco_c32* Z, int64_t n_sig, float* out, disco_stream s) {
    DISCO_ENTER(ctx);
    return istft_any(ctx, Z, n_sig, out, ctx->cfg.length, ctx->T, s, false);
}
