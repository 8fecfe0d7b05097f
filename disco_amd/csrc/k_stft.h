// STFT / oracle-mask / iSTFT kernels (librosa semantics, see oracle/stft_oracle.py for the restated
// algorithm and the reference call sites: tango.py:335-342, 528-539; math_utils.py:134-152).
//
// A wave owns one frame of one signal group at a time and transforms its channels two per complex FFT
// (fft.h).  Window and twiddle factors depend on the lane only and are held in registers; the only LDS is the
// wave-private exchange buffer, so the four waves of a block never synchronise with each other in the
// forward kernels.  All samples of a frame are requested before the first butterfly, and the spectra of all
// channel pairs are written together: one contiguous 8*chans-byte vector per (frame, bin), i.e. 512*chans
// contiguous bytes per wave store.
#pragma once
#include "fft.h"
#include "k_cov.h"

namespace disco {

#ifndef DISCO_STFT_WAVES
#define DISCO_STFT_WAVES 4
#endif
#ifndef DISCO_STFT_RUN
#define DISCO_STFT_RUN 16
#endif
constexpr int STFT_WAVES = DISCO_STFT_WAVES;     // waves per block; each streams its own run of frames
constexpr int STFT_RUN = DISCO_STFT_RUN;         // consecutive frames per wave
// host side: launch geometry of the per-wave-run kernels
inline int stft_runs(int T) { return (T + STFT_RUN - 1) / STFT_RUN; }
inline long long stft_blocks(long long n_witems) { return (n_witems + STFT_WAVES - 1) / STFT_WAVES; }

template <int N>
struct StftShared {
    c32 buf[STFT_WAVES][fft_buf_len<N>()];
};

template <int N>
__device__ __forceinline__ void load_window(float* w, const float* __restrict__ win, int lane) {
#pragma unroll
    for (int e = 0; e < FftPlan<N>::E; ++e) w[e] = win[lane + 64 * e];
}
// The analysis window at HALF weight: the transforms below run on (a + i b) / 2, which is what lets rfft_pair_untangle
// separate the two spectra with one packed add each (fft.h).  0.5 w is exact, so every spectrum is bit-for-bit what the
// full-weight window followed by the division gives.
template <int N>
__device__ __forceinline__ void load_window_half(float* w, const float* __restrict__ win, int lane) {
#pragma unroll
    for (int e = 0; e < FftPlan<N>::E; ++e) w[e] = 0.5f * win[lane + 64 * e];
}
// (a, b) * (w, w) for a full channel pair, (a, 0) * w when the pair's second channel does not exist.  One packed multiply per
// slot: the window registers are read as pairs (w[2m], w[2m+1]) and the slot's weight is broadcast by the operand selectors.
template <int N>
__device__ __forceinline__ void apply_window(c32* v, const c32* raw, const float* w, bool has_b) {
#pragma unroll
    for (int m = 0; m < FftPlan<N>::E / 2; ++m) {
        const c32 wp = make_float2(w[2 * m], w[2 * m + 1]);
        v[2 * m] = scale_by_half<0>(raw[2 * m], wp);
        v[2 * m + 1] = scale_by_half<1>(raw[2 * m + 1], wp);
        if (!has_b) {
            v[2 * m].y = 0.f;
            v[2 * m + 1].y = 0.f;
        }
    }
}

// Slots [E0, E1) of centre-padded frame t of the channel pair (xa, xb), un-windowed: v[e] = (a, b)[lane + 64 e].
// With hop = N/2 the first half of frame t+1 is the second half of frame t, so a wave streaming consecutive
// frames only ever fetches slots [E/2, E) after its first frame.
template <int N, int E0, int E1>
__device__ __forceinline__ void load_frame_slots(c32* v, const float* __restrict__ xa, const float* __restrict__ xb,
                                                 int t, int L, int pad_mode, int lane) {
    constexpr int H = N / 2;
    const int p0 = t * H - N / 2;
    const bool interior = (p0 + 64 * E0 >= 0) && (p0 + 64 * E1 <= L);      // wave-uniform (t is)
    // xb == xa for an odd channel count: the caller zeroes that half through the window (wb), so the loaded
    // values are never touched here -- a prefetched half-window must stay un-consumed until after the stores of the
    // current frame are issued (vmcnt is in-order: consuming a load early also waits for every older store).
    if (interior) {
#pragma unroll
        for (int e = E0; e < E1; ++e) {
            const int n = lane + 64 * e;
            v[e - E0] = make_float2(xa[p0 + n], xb[p0 + n]);
        }
    } else {                                                               // first / last frames only
#pragma unroll
        for (int e = E0; e < E1; ++e) {
            const int n = lane + 64 * e;
            v[e - E0] = make_float2(load_padded(xa, p0 + n, L, pad_mode), load_padded(xb, p0 + n, L, pad_mode));
        }
    }
}

// x: [n_sig][chans][L] -> X: [n_sig][T][F][chans].  CHP = ceil(chans / 2) channel pairs per frame.
// A wave streams STFT_RUN consecutive frames of one signal group: the next frame's new half-window is
// requested before the current frame is transformed (the loads fly under the butterflies), the shared
// half-window is recycled in registers.
template <int N, int CHP>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * STFT_WAVES) void k_stft(const float* __restrict__ x, c32* __restrict__ X,
                                                           const float* __restrict__ win, const c32* __restrict__ tw,
                                                           int chans, int L, int T, int pad_mode, int runs_per_sig,
                                                           long long n_witems) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1, NJ = E / 2 + 1, EH = E / 2;
    __shared__ StftShared<N> sh;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * STFT_WAVES + wave;          // (g, run)
    if (item >= n_witems) return;                  // no block-level synchronisation anywhere below
    const long long g = item / runs_per_sig;
    const int t0 = (int)(item % runs_per_sig) * STFT_RUN;
    const int t1 = min(T, t0 + STFT_RUN);
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    float w[E];
    load_window_half<N>(w, win, lane);
    const float* xa[CHP];
    const float* xb[CHP];
#pragma unroll
    for (int p = 0; p < CHP; ++p) {
        xa[p] = x + (g * chans + 2 * p) * (long long)L;
        xb[p] = (2 * p + 1 < chans) ? x + (g * chans + 2 * p + 1) * (long long)L : xa[p];
    }
    c32 raw[CHP][E];
#pragma unroll
    for (int p = 0; p < CHP; ++p) load_frame_slots<N, 0, E>(raw[p], xa[p], xb[p], t0, L, pad_mode, lane);
    for (int t = t0; t < t1; ++t) {
        c32 nxt[CHP][EH];
        {
            const int tn = min(t + 1, T - 1);              // clamped: harmless reload at the end of a run
#pragma unroll
            for (int p = 0; p < CHP; ++p) load_frame_slots<N, EH, E>(nxt[p], xa[p], xb[p], tn, L, pad_mode, lane);
        }
        c32 A[CHP][NJ], B[CHP][NJ];
#pragma unroll
        for (int p = 0; p < CHP; ++p) {
            c32 v[E];
            apply_window<N>(v, raw[p], w, 2 * p + 1 < chans);
            fft_wave<N>(v, wtw, sh.buf[wave], lane);
            rfft_pair_untangle<N>(v, sh.buf[wave], lane, [&](int j, int, c32 a, c32 b) {
                A[p][j] = a;
                B[p][j] = b;
            });
        }
        // Consume the prefetched half-window BEFORE this frame's stores are issued: vmcnt retires in order, so a wait
        // placed after the stores would also wait for them; placed here it only covers loads (and stores of the
        // previous frame, a whole transform old).
#pragma unroll
        for (int p = 0; p < CHP; ++p)
#pragma unroll
            for (int e = 0; e < EH; ++e) {
                DISCO_CONSUME(nxt[p][e].x);
                DISCO_CONSUME(nxt[p][e].y);
                raw[p][e] = raw[p][e + EH];
                raw[p][e + EH] = nxt[p][e];
            }
        {
            c32* Xo = X + ((g * T + t) * (long long)F) * chans;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j < NJ - 1 || lane == 0) {
                    const int f = lane + 64 * j;
                    c32* o = Xo + (long long)f * chans;
                    if ((chans & 1) == 0) {
                        float4* o4 = reinterpret_cast<float4*>(o);      // (f*chans + 2p) * 8 B is 16-B aligned
#pragma unroll
                        for (int p = 0; p < CHP; ++p) o4[p] = make_float4(A[p][j].x, A[p][j].y, B[p][j].x, B[p][j].y);
                    } else {
#pragma unroll
                        for (int p = 0; p < CHP; ++p) {
                            o[2 * p] = A[p][j];
                            if (2 * p + 1 < chans) o[2 * p + 1] = B[p][j];
                        }
                    }
                }
            }
        }
    }
}

// More than 4 channels (and N = 1024 from 3 on): k_stft<N, CHP> keeps every channel pair's window AND spectrum of a frame in
// registers (N = 1024, CHP = 4: 256 VGPRs + 214 AGPR copies, one wave per SIMD -- 7.5 ms on the C5 batch).  Here the waves of a
// workgroup split the pairs instead: wave p transforms channel pair p of the same (signal group, run of frames) and parks its
// two spectra in column p of an LDS tile laid out exactly like the frame's [F][chans] block of X; after a barrier the whole
// workgroup copies the tile out linearly (1 KiB of consecutive bytes per wave store; storing each pair's 16 bytes per bin
// straight from its wave, at a pitch of chans * 8, measured 40 % SLOWER than k_stft: four partial writes per 64-byte line).
template <int N>
struct alignas(16) StftPairsShared {
    c32 buf[STFT_WAVES][fft_buf_len<N>()];
    c32 tile[N / 2 + 1][2 * STFT_WAVES];
};
// Where the 16-byte granule (bin f, channel pair p) of the tile lives, in granules.  Lane = bin, so a wave's ds_write_b128 puts
// 16 consecutive bins at a pitch of chp granules: with chp = 4 (or 2) only 4 (8) of the 16 granule slots of a 256-byte LDS row
// are hit -- a 4-way (2-way) bank conflict on every store (round 2's counters: 41 % of the kernel's LDS cycles).  XOR-ing the
// pair index with bits of the bin spreads the 16 bins over all 16 slots; the copy-out undoes it with the same expression, and
// its 16 consecutive granules still cover 16 different slots.
__device__ __forceinline__ int pairs_tile_slot(int f, int p, int chp) {
    const int sw = chp == 4 ? ((f >> 2) & 3) : (chp == 2 ? ((f >> 3) & 1) : 0);
    return f * chp + (p ^ sw);
}
// the same for a linear granule index i = f * chp + p (the copy-out's): no division
__device__ __forceinline__ int pairs_tile_slot_linear(int i, int chp) {
    return chp == 4 ? (i ^ ((i >> 4) & 3)) : (chp == 2 ? (i ^ ((i >> 4) & 1)) : i);
}

template <int N>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * STFT_WAVES, 2) void k_stft_pairs(const float* __restrict__ x, c32* __restrict__ X,
                                                                    const float* __restrict__ win, const c32* __restrict__ tw,
                                                                    int chans, int L, int T, int pad_mode, int runs_per_sig) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1, NJ = E / 2 + 1, EH = E / 2;
    __shared__ StftPairsShared<N> sh;
    const int wave = wave_id(), lane = threadIdx.x & 63, tid = threadIdx.x;
    const bool mine = 2 * wave < chans;            // waves without a pair still take part in the barriers and the copy
    const bool two = 2 * wave + 1 < chans;
    const int chp = (chans + 1) / 2;
    const long long g = blockIdx.x / runs_per_sig;
    const int t0 = (int)(blockIdx.x % runs_per_sig) * STFT_RUN;
    const int t1 = min(T, t0 + STFT_RUN);
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    float w[E];
    load_window_half<N>(w, win, lane);
    const float* xa = x + (g * chans + (mine ? 2 * wave : 0)) * (long long)L;
    const float* xb = two ? xa + L : xa;
    c32 raw[E];
    load_frame_slots<N, 0, E>(raw, xa, xb, t0, L, pad_mode, lane);
    c32* tile = &sh.tile[0][0];                    // row pitch 2 * chp: the frame's X block when chans is even
    for (int t = t0; t < t1; ++t) {
        c32 nxt[EH];
        load_frame_slots<N, EH, E>(nxt, xa, xb, min(t + 1, T - 1), L, pad_mode, lane);
        if (mine) {
            c32 v[E];
            apply_window<N>(v, raw, w, two);
            fft_wave<N>(v, wtw, sh.buf[wave], lane);
            rfft_pair_untangle<N>(v, sh.buf[wave], lane, [&](int, int f, c32 a, c32 b) {
                *reinterpret_cast<float4*>(&tile[pairs_tile_slot(f, wave, chp) * 2]) = make_float4(a.x, a.y, b.x, b.y);
            });
        }
#pragma unroll
        for (int e = 0; e < EH; ++e) {                 // consume the prefetch before the stores (see k_stft)
            DISCO_CONSUME(nxt[e].x);
            DISCO_CONSUME(nxt[e].y);
            raw[e] = raw[e + EH];
            raw[e + EH] = nxt[e];
        }
        __syncthreads();
        c32* Xo = X + ((g * T + t) * (long long)F) * chans;
        if ((chans & 1) == 0) {
            const float4* src = reinterpret_cast<const float4*>(tile);
            float4* dst = reinterpret_cast<float4*>(Xo);
            const int shift = (int)((reinterpret_cast<unsigned long long>(dst) >> 4) & 7);      // whole 128-byte lines per wave store (see k_stft_cov)
            for (int i = tid - shift; i < F * chp; i += 64 * STFT_WAVES)
                if (i >= 0) store_stream16(&dst[i], src[pairs_tile_slot_linear(i, chp)]);
        } else {
            for (int i = tid; i < F * chans; i += 64 * STFT_WAVES) {
                const int f = i / chans, c = i % chans;
                Xo[i] = tile[pairs_tile_slot(f, c >> 1, chp) * 2 + (c & 1)];
            }
        }
        __syncthreads();                               // the tile is rewritten by the next frame
    }
}

// tf_mask (dnn/utils.py:57-67) on one bin.  v_sqrt_f32 / v_rcp_f32 (1 ulp) instead of the IEEE div/sqrt expansions:
// ~12 instructions instead of ~60 per bin, error 2-3 ulp on a mask that is compared at 1e-5.
__device__ __forceinline__ float ipow(float r, int p) {
    float m = r;
    for (int i = 1; i < p; ++i) m *= r;
    return p == 0 ? 1.f : m;
}
__device__ __forceinline__ float tf_mask_value(c32 S, c32 Nn, int mask_type, int mask_pow, float thr_lin) {
    const float as = __builtin_amdgcn_sqrtf(S.x * S.x + S.y * S.y);
    if (mask_type == DISCO_MASK_IAM) {
        const c32 y = cadd(S, Nn);
        return ipow(as * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(y.x * y.x + y.y * y.y)), mask_pow);
    }
    const float an = fmaxf(__builtin_amdgcn_sqrtf(Nn.x * Nn.x + Nn.y * Nn.y), 2.220446049250313e-16f);
    const float xi = ipow(as * __builtin_amdgcn_rcpf(an), mask_pow);
    if (mask_type == DISCO_MASK_IBM) return xi >= thr_lin ? 1.f : 0.f;
    // xi / (1 + xi); an overflowing xi gives NaN exactly like the reference's inf / inf
    return xi * __builtin_amdgcn_rcpf(1.f + xi);
}

// s_ref, n_ref: [n_sig][L] -> mask [n_sig][T][F]; the pair (s, n) shares one complex FFT.  Streams like k_stft.
template <int N>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * STFT_WAVES) void k_mask_oracle(const float* __restrict__ s_ref, const float* __restrict__ n_ref,
                                                                  float* __restrict__ mask, const float* __restrict__ win,
                                                                  const c32* __restrict__ tw, int L, int T, int pad_mode,
                                                                  int mask_type, int mask_pow, float thr_lin, int runs_per_sig,
                                                                  long long n_witems) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1, EH = E / 2;
    __shared__ StftShared<N> sh;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * STFT_WAVES + wave;
    if (item >= n_witems) return;
    const long long g = item / runs_per_sig;
    const int t0 = (int)(item % runs_per_sig) * STFT_RUN;
    const int t1 = min(T, t0 + STFT_RUN);
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    float w[E];
    load_window_half<N>(w, win, lane);
    const float* xs = s_ref + g * (long long)L;
    const float* xn = n_ref + g * (long long)L;
    c32 raw[E];
    load_frame_slots<N, 0, E>(raw, xs, xn, t0, L, pad_mode, lane);
    for (int t = t0; t < t1; ++t) {
        c32 nxt[EH];
        load_frame_slots<N, EH, E>(nxt, xs, xn, min(t + 1, T - 1), L, pad_mode, lane);
        c32 v[E];
        apply_window<N>(v, raw, w, true);
        fft_wave<N>(v, wtw, sh.buf[wave], lane);
        float mval[EH + 1];
        rfft_pair_untangle<N>(v, sh.buf[wave], lane, [&](int j, int, c32 a, c32 b) {
            mval[j] = tf_mask_value(a, b, mask_type, mask_pow, thr_lin);
        });
#pragma unroll
        for (int e = 0; e < EH; ++e) {                 // consume the prefetch before the stores (see k_stft)
            DISCO_CONSUME(nxt[e].x);
            DISCO_CONSUME(nxt[e].y);
            raw[e] = raw[e + EH];
            raw[e + EH] = nxt[e];
        }
        float* mo = mask + (g * T + t) * (long long)F;
#pragma unroll
        for (int j = 0; j <= EH; ++j)
            if (j < EH || lane == 0) store_stream4(&mo[lane + 64 * j], mval[j]);
    }
}

static __global__ void k_tf_mask(const c32* __restrict__ S, const c32* __restrict__ Nn, float* __restrict__ mask,
                          long long n, int mask_type, int mask_pow, float thr_lin) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        mask[i] = tf_mask_value(S[i], Nn[i], mask_type, mask_pow, thr_lin);
}

static __global__ void k_tf_mask_channel(const c32* __restrict__ S, const c32* __restrict__ Nn, float* __restrict__ mask, long long n, int M,
                                  int ch, int mask_type, int mask_pow, float thr_lin) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        mask[i] = tf_mask_value(S[i * M + ch], Nn[i * M + ch], mask_type, mask_pow, thr_lin);
}

// ---- iSTFT --------------------------------------------------------------------------------------------
// Z: [n_sig][T][F] -> out: [n_sig][L].  A block owns ISTFT_FRAMES consecutive frames of one signal (each wave
// inverts two frames with one complex FFT), overlap-adds them in LDS and emits the ISTFT_FRAMES - 1 hop
// segments that are complete inside the block; consecutive blocks overlap by one frame.
constexpr int ISTFT_FRAMES = 2 * STFT_WAVES;
constexpr int ISTFT_SEGS = ISTFT_FRAMES - 1;

template <int N>
struct IstftShared {
    float win[N];
    c32 buf[STFT_WAVES][fft_buf_len<N>()];     // after the transform each wave parks its two time frames here
};
template <int N>
__device__ __forceinline__ float* istft_frame(IstftShared<N>& sh, int j) {
    return reinterpret_cast<float*>(sh.buf[j >> 1]) + (j & 1) * N;
}

// SOLO: every wave inverts ONE frame per transform (the second slot of the pair stays empty) and a block owns STFT_WAVES frames.  Twice the
// transforms -- but a frame's samples then depend on its own spectrum only, not at rounding level on the partner it shared a transform
// with, which is what lets the STREAMING online path (disco_tango_online_stream) reproduce the whole-clip call bit for bit whatever the
// chunking: both online entry points invert this way (their overlap-add is 1 % of their time).
template <int N, bool SOLO = false>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * STFT_WAVES) void k_istft(const c32* __restrict__ Z, float* __restrict__ out,
                                                            const float* __restrict__ win, const c32* __restrict__ tw,
                                                            int L, int T, int blocks_per_sig) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1, H = N / 2;
    constexpr int SEGS = SOLO ? STFT_WAVES - 1 : ISTFT_SEGS;
    __shared__ IstftShared<N> sh;
    for (int i = threadIdx.x; i < N; i += blockDim.x) sh.win[i] = win[i];
    const int wave = wave_id(), lane = threadIdx.x & 63;
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    const long long g = blockIdx.x / blocks_per_sig;
    const int seg0 = (int)(blockIdx.x % blocks_per_sig) * SEGS;             // first output segment == first frame
    const int ta = SOLO ? seg0 + wave : seg0 + 2 * wave, tb = ta + 1;
    const c32* Za = Z + (g * T + ta) * (long long)F;
    const c32* Zb = Z + (g * T + tb) * (long long)F;
    const bool has_a = ta < T, has_b = !SOLO && tb < T;
    // V[n] = A~[n] + i B~[n] with A~, B~ the Hermitian extensions; the inverse transform is conj(FFT(conj V)) / N
    c32 v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int n = lane + 64 * e;
        const bool upper = n > N / 2;
        const int f = upper ? N - n : n;
        c32 a = has_a ? Za[f] : make_float2(0.f, 0.f);
        c32 b = has_b ? Zb[f] : make_float2(0.f, 0.f);
        if (f == 0 || f == N / 2) {      // irfft ignores the imaginary part of DC and Nyquist
            a.y = 0.f;
            b.y = 0.f;
        }
        if (upper) {
            a.y = -a.y;
            b.y = -b.y;
        }
        const c32 V = make_float2(a.x - b.y, a.y + b.x);
        v[e] = cconj(V);
    }
    fft_wave<N>(v, wtw, sh.buf[wave], lane);
    const float inv = 1.0f / N;
    __syncthreads();                                   // sh.win is loaded; every lane is past its last read of buf[wave]
    // frame slot of block-local frame j: SOLO keeps one frame per wave buffer, else two
    auto frame = [&](int j) { return SOLO ? reinterpret_cast<float*>(sh.buf[j]) : istft_frame<N>(sh, j); };
    float* fa = frame(SOLO ? wave : 2 * wave);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int n = lane + 64 * e;
        const float w = sh.win[n] * inv;
        fa[n] = v[e].x * w;                            // Re(conj(out)) =  out.x
        if constexpr (!SOLO) frame(2 * wave + 1)[n] = -v[e].y * w;       // Im(conj(out)) = -out.y
    }
    __syncthreads();
    float* o = out + g * (long long)L;
    for (int i = threadIdx.x; i < SEGS * H; i += blockDim.x) {
        const int j = i / H, n = i - j * H;
        const int seg = seg0 + j;
        const long long pos = (long long)seg * H + n;
        if (seg < T && pos < L) {
            const float w0 = sh.win[H + n], w1 = sh.win[n];
            float wss = w0 * w0;
            if (seg + 1 < T) wss += w1 * w1;
            float val = frame(j)[H + n] + frame(j + 1)[n];
            if (wss > 1.17549435e-38f) val /= wss;
            o[pos] = val;
        }
    }
}

}  // namespace disco

namespace disco {

// ---- STFT + step-1 covariance in one pass over the samples ---------------------------------------------------
// tango.py:335 + 357-364: the spectra are needed twice right away (stored for step 2, and reduced into the local
// covariances), so the 4 waves of a workgroup stream 4 interleaved runs of frames of one node, park each
// frame's M-channel spectrum in an LDS tile, and after a barrier the workgroup switches to one-thread-per-bin:
// every thread copies its bin's M-vector to X (coalesced 8*M*64-byte wave stores) and folds it into the
// P(P+1)/2 covariance accumulators it keeps in registers for the whole run.  Saves the separate covariance
// pass over X (8*M*F bytes per node-frame).
// frames per wave per workgroup are a launch parameter (`runw`; a workgroup covers 4 * runw frames): long runs amortise the
// first-frame load and the partial-sum write-out (80 is best at C3), short ones keep small batches spread over the chip

// ZPITCH: pitch of a pair plane in the Z layout (DISCO_ZTILE below): N + 16 complex, i.e. consecutive planes start 32 banks apart, so the
// copy-out's ds_read_b64 -- consecutive lanes alternate between the planes of one bin -- never meet on a bank
template <int N>
constexpr int stft_cov_zpitch() { return N + 16; }
template <int N, int CHP, bool ZT>
struct alignas(16) StftCovShared {
    c32 buf[STFT_WAVES][fft_buf_len<N>()];
    c32 tile[STFT_WAVES][ZT ? CHP * stft_cov_zpitch<N>() : (N / 2 + 1) * 2 * CHP];
};

// 3 waves per SIMD (<= 168 VGPRs) is reachable for the common 512-point, M <= 4 shape and worth ~3 %; larger shapes keep
// whatever occupancy their register need allows
#ifndef DISCO_SC_WPE
#define DISCO_SC_WPE 3
#endif
// DISCO_ZTILE: the transform waves park the raw pair spectra Z (natural order, one conflict-free ds_write_b64 per slot) and whoever picks
// a bin up separates the two channels himself -- A = Z[f] + conj Z[N - f], B = -i (Z[f] - conj Z[N - f]): the very operations of
// rfft_pair_untangle on the very operands, so every spectrum and every sum is bit for bit what the untangled tile gives.  What it saves
// sits on the LDS pipe, which together with the VALU bounds the STORE = false variant (profiles/r03_q_C2x4000_pmc_alu.json: ~48 % busy
// each, barely overlapped) -- per transform the 2 E ds_bpermute of the untangle (an LDS store + load each) and E/2 ds_write_b128 at a
// 32-byte pitch (13 cycles each, MI355X_MICROARCH.md "LDS") against E ds_write_b64 (6 each): ~280 -> ~180 LDS cycles per transform --
// and the untangle's registers in the transform phase, the phase that sets this kernel's register count.
#ifndef DISCO_ZTILE
#define DISCO_ZTILE 1
#endif
// DISCO_SC_EXP (default 0): TIMING-ONLY builds of k_stft_cov with parts of the frame loop removed -- bit 0 the transforms (window, FFT, tile
// write), 1 the covariance fold, 2 the X store, 3 the sample and mask loads; results are garbage, only the time counts
// (tools/gpu/mk_variant.sh scexp<N> "-DDISCO_SC_EXP=<N>" api_stft_cov; profiles/r05_h_stft_parts.txt)
#ifndef DISCO_SC_EXP
#define DISCO_SC_EXP 0
#endif
// STORE = false: the spectra are reduced into the covariances and dropped (single-node path: the filter pass recomputes them
// from the samples, k_stft_apply_istft, instead of reading 8 M F bytes per node-frame back)
template <int N, int M, bool STORE = true>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * STFT_WAVES, (N == 512 && M <= 4) ? DISCO_SC_WPE : 1) void k_stft_cov(const float* __restrict__ x, const float* __restrict__ mask,
                                                               c32* __restrict__ X, float4* __restrict__ part,
                                                               const float* __restrict__ win, const c32* __restrict__ tw,
                                                               int L, int T, int pad_mode, int chunks, int runw) {
    static_assert(STFT_WAVES == 4, "one bin per thread needs 4 waves for 256 bins");
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1, EH = E / 2, CHP = (M + 1) / 2, MP = 2 * CHP;
    constexpr int NP = M * (M + 1) / 2;
    constexpr int BPT = (F - 1) / 256;             // bins per thread: 1 (N = 512) or 2 (N = 1024)
    constexpr bool ZT = DISCO_ZTILE != 0;
    constexpr int ZP = stft_cov_zpitch<N>();
    __shared__ StftCovShared<N, CHP, ZT> sh;
    // X layout: row (frame of wave ww, bin f) = the 2 CHP channels as they go to HBM; Z layout: plane (ww, pair p) = Z[0 .. N)
    auto xrow = [&](int ww, int f) { return &sh.tile[ww][f * 2 * CHP]; };
    auto zplane = [&](int ww, int p) { return &sh.tile[ww][p * ZP]; };
    const int tid = threadIdx.x, wave = wave_id(), lane = tid & 63;
    const long long g = blockIdx.x / chunks;
    const int c = (int)(blockIdx.x % chunks);
    const int tb = c * STFT_WAVES * runw;
    const int ts = tb + wave * runw;
    const int te = min(T, ts + runw);
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    float w[E];
    load_window_half<N>(w, win, lane);
    const float* xa[CHP];
    const float* xb[CHP];
#pragma unroll
    for (int p = 0; p < CHP; ++p) {
        xa[p] = x + (g * M + 2 * p) * (long long)L;
        xb[p] = (2 * p + 1 < M) ? x + (g * M + 2 * p + 1) * (long long)L : xa[p];
    }
    c32 raw[CHP][E];
#pragma unroll
    for (int p = 0; p < CHP; ++p) {
        if (ts < te) load_frame_slots<N, 0, E>(raw[p], xa[p], xb[p], ts, L, pad_mode, lane);
        else {
#pragma unroll
            for (int e = 0; e < E; ++e) raw[p][e] = make_float2(0.f, 0.f);
        }
    }
    c32 acc_s[BPT][NP], acc_n[BPT][NP];
#pragma unroll
    for (int b = 0; b < BPT; ++b)
#pragma unroll
        for (int q = 0; q < NP; ++q) acc_s[b][q] = acc_n[b][q] = make_float2(0.f, 0.f);
    // Nyquist bin: thread (q, which) with tid < 2*NP owns one complex entry of Rss (which = 0) or Rnn (1)
    c32 acc_ny = make_float2(0.f, 0.f);
    int ny_i = 0, ny_j = 0;
    {
        int q = tid >> 1, i = 0;
        while (i < M - 1 && q >= M - i) {
            q -= M - i;
            ++i;
        }
        ny_i = i;
        ny_j = i + q;
    }
    const float* mg = mask + g * T * (long long)F;
    for (int it = 0; it < runw; ++it) {
        const int t = ts + it;
        const bool valid = t < te;
        // masks of the 4 frames this thread will reduce after the barrier (requested early)
        float mv[STFT_WAVES][BPT], mny[STFT_WAVES];
#pragma unroll
        for (int ww = 0; ww < STFT_WAVES; ++ww) {
            const int t2 = min(tb + ww * runw + it, T - 1);          // clamped: unconditional loads, used only when valid
#pragma unroll
            for (int b = 0; b < BPT; ++b) mv[ww][b] = (DISCO_SC_EXP & 8) ? 0.5f : mg[(long long)t2 * F + tid + 256 * b];
            mny[ww] = (DISCO_SC_EXP & 8) ? 0.5f : mg[(long long)t2 * F + F - 1];
        }
        c32 nxt[CHP][EH];
        {
            const int tn = min(t + 1, T - 1);              // clamped: harmless reload at the end of a run
#pragma unroll
            for (int p = 0; p < CHP; ++p) {
                if (DISCO_SC_EXP & 8) {
#pragma unroll
                    for (int e = 0; e < EH; ++e) nxt[p][e] = make_float2(0.25f * e + lane, 1.f);
                } else {
                    load_frame_slots<N, EH, E>(nxt[p], xa[p], xb[p], tn, L, pad_mode, lane);
                }
            }
        }
        if (valid && !(DISCO_SC_EXP & 1)) {
#pragma unroll
            for (int p = 0; p < CHP; ++p) {
                c32 v[E];
                apply_window<N>(v, raw[p], w, 2 * p + 1 < M);
                fft_wave<N>(v, wtw, sh.buf[wave], lane);
                if constexpr (ZT) {
                    c32* zp = zplane(wave, p);
#pragma unroll
                    for (int e = 0; e < E; ++e) zp[lane + 64 * e] = v[e];
                } else {
                    // (NOT swizzled like k_stft_pairs' tile: measured, the swizzle costs this kernel 8 % -- 7.70 against 7.10 ms per C3 launch;
                    // its stores are 2-way conflicted at worst and the copy-out below wants the plain linear read)
                    rfft_pair_untangle<N>(v, sh.buf[wave], lane, [&](int, int f, c32 a, c32 b) {
                        *reinterpret_cast<float4*>(xrow(wave, f) + 2 * p) = make_float4(a.x, a.y, b.x, b.y);
                    });
                }
            }
        }
#pragma unroll
        for (int p = 0; p < CHP; ++p)                   // consume the prefetch before the X stores below (see k_stft)
#pragma unroll
            for (int e = 0; e < EH; ++e) {
                DISCO_CONSUME(nxt[p][e].x);
                DISCO_CONSUME(nxt[p][e].y);
                raw[p][e] = raw[p][e + EH];
                raw[p][e + EH] = nxt[p][e];
            }
        __syncthreads();
        // ---- one thread per bin: copy out + reduce the (up to) 4 frames of this iteration
#pragma unroll
        for (int ww = 0; ww < STFT_WAVES; ++ww) {
            const int t2 = tb + ww * runw + it;
            if (t2 < min(T, tb + (ww + 1) * runw)) {          // workgroup-uniform
                c32* Xo = STORE ? X + ((g * T + t2) * (long long)F) * M : nullptr;
                if (STORE && (M & 1) == 0 && !(DISCO_SC_EXP & 4)) {
                    // even M: the tile row IS the X row (F*M complex, contiguous) -> straight 16-B-per-lane copy, every
                    // wave store covers 1 KiB of consecutive bytes (a per-bin store would touch each 128-B line twice)
                    // The rows are 8 * M * F bytes long (8224 for M = 4), so they start 0 / 32 / 64 / 96 bytes into a 128-byte line: the
                    // copy is shifted by that much, every wave store then covers whole lines (PMC: stores that straddle lines
                    // at both ends cost a fill read per partial line, +2.8 GB of reads per C3 launch).
                    float4* dst = reinterpret_cast<float4*>(Xo);
                    const int shift = (int)((reinterpret_cast<unsigned long long>(dst) >> 4) & 7);
                    if constexpr (ZT) {             // granule i = (bin i / CHP, pair i % CHP): untangled on the way out
                        for (int i = tid - shift; i < F * CHP; i += 64 * STFT_WAVES)
                            if (i >= 0) {
                                const int f = i / CHP, pp = i - f * CHP;
                                const c32* zp = zplane(ww, pp);
                                const c32 z = zp[f], zc = zp[(N - f) & (N - 1)];
                                const c32 a = cadd_conj(z, zc), b = csub_conj_mi(z, zc);
                                store_stream16(&dst[i], make_float4(a.x, a.y, b.x, b.y));
                            }
                    } else {
                        const float4* src = reinterpret_cast<const float4*>(xrow(ww, 0));
                        for (int i = tid - shift; i < F * M / 2; i += 64 * STFT_WAVES)
                            if (i >= 0) store_stream16(&dst[i], src[i]);
                    }
                }
#pragma unroll
                for (int b = 0; b < BPT; ++b) {
                    const int f = tid + 256 * b;
                    c32 xv[MP];
#pragma unroll
                    for (int p = 0; p < CHP; ++p) {
                        if constexpr (ZT) {
                            const c32* zp = zplane(ww, p);
                            const c32 z = zp[f], zc = zp[(N - f) & (N - 1)];
                            xv[2 * p] = cadd_conj(z, zc);
                            xv[2 * p + 1] = csub_conj_mi(z, zc);
                        } else {
                            const float4 q4 = *reinterpret_cast<const float4*>(xrow(ww, f) + 2 * p);
                            xv[2 * p] = make_float2(q4.x, q4.y);
                            xv[2 * p + 1] = make_float2(q4.z, q4.w);
                        }
                    }
                    if (STORE && (M & 1) != 0) {
#pragma unroll
                        for (int i = 0; i < M; ++i) Xo[(long long)f * M + i] = xv[i];
                    }
                    const float m = mv[ww][b], mc = 1.f - m;
                    if (!(DISCO_SC_EXP & 2)) cov_accumulate_shared<M>(xv, m * m, mc * mc, acc_s[b], acc_n[b]);
                    else acc_s[b][0].x += xv[0].x * m;           // (keeps the tile reads alive)
                }
                // Nyquist bin
                if (STORE && (M & 1) != 0 && tid < M) {
                    if constexpr (ZT) {
                        const c32 zq = zplane(ww, tid >> 1)[N / 2];
                        Xo[(long long)(F - 1) * M + tid] = (tid & 1) ? csub_conj_mi(zq, zq) : cadd_conj(zq, zq);
                    } else {
                        Xo[(long long)(F - 1) * M + tid] = xrow(ww, F - 1)[tid];
                    }
                }
                if (tid < 2 * NP) {
                    const float m = (tid & 1) ? 1.f - mny[ww] : mny[ww];
                    c32 a, b2;
                    if constexpr (ZT) {             // channel i of the Nyquist bin from its pair's Z[N / 2] (its own partner)
                        const c32 zi = zplane(ww, ny_i >> 1)[N / 2], zj = zplane(ww, ny_j >> 1)[N / 2];
                        a = (ny_i & 1) ? csub_conj_mi(zi, zi) : cadd_conj(zi, zi);
                        b2 = (ny_j & 1) ? csub_conj_mi(zj, zj) : cadd_conj(zj, zj);
                    } else {
                        a = xrow(ww, F - 1)[ny_i];
                        b2 = xrow(ww, F - 1)[ny_j];
                    }
                    const float m2 = m * m;
                    acc_ny.x = fmaf(m2, a.x * b2.x + a.y * b2.y, acc_ny.x);
                    if (ny_i != ny_j) acc_ny.y = fmaf(m2, a.y * b2.x - a.x * b2.y, acc_ny.y);
                }
            }
        }
        __syncthreads();
    }
    float4* o = part + ((g * chunks + c) * F) * (long long)NP;
#pragma unroll
    for (int b = 0; b < BPT; ++b)
#pragma unroll
        for (int q = 0; q < NP; ++q)
            o[(long long)(tid + 256 * b) * NP + q] = make_float4(acc_s[b][q].x, acc_s[b][q].y, acc_n[b][q].x, acc_n[b][q].y);
    if (tid < 2 * NP) {
        float2* o2 = reinterpret_cast<float2*>(o + (long long)(F - 1) * NP + (tid >> 1));
        o2[tid & 1] = acc_ny;
    }
}

}  // namespace disco
