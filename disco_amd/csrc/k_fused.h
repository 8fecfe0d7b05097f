// Step-2 kernels with the cross-node z exchange kept ON CHIP (mask_for_z = 'local', tango.py:378-450).
//
// DISCO's algorithm has every node k send one compressed channel z_k = w_loc,k^H y_k to all the others
// (tango.py:369, 382).  When all K nodes of a room live on one GPU that exchange never needs to touch HBM:
// a workgroup has one WAVE per node; wave k streams node k's STFT for a tile of 64 bins (one contiguous
// 512*M-byte span per wave-load), filters it with its local filter held in registers, drops its z into a
// double-buffered LDS tile, and after one workgroup barrier per U frames picks up the K-1 remote z's.
//   k_step2_cov_fused   : z (optionally written out) + the (M+K-1)x(M+K-1) masked covariances of every node
//                         -- replaces disco_apply + disco_cov_masked(Zs=Zn=z): two passes over X and four over z less
//   k_step2_apply_fused : z again (recomputed, 4 FMAs per channel) + yf_k = w_glo,k^H [y_k ; z_-k]
// The Nyquist bin (F = 64 n + 1) is served by one extra workgroup per (room, chunk) whose lanes stride over
// frames instead of bins; the exchange is identical (same lane <-> same frame in every wave).
#pragma once
#include "k_cov.h"

namespace disco {

struct Step2Args {
    const c32* X;        // [R][K][T][F][M]
    const float* mask;   // [R][K][T][F]         (cov kernel)
    const c32* w_loc;    // [R][K][F][M]
    const c32* w_glo;    // [R][K][F][M+K-1]     (apply kernel)
    c32* z_out;          // [R][K][T][F] or null
    c32* yf;             // [R][K][T][F]         (apply kernel)
    float4* part;        // [R*K][chunks][F][NP] (cov kernel)
    int K, T, F, chunks;
};

// (room, bin tile | Nyquist, frame chunk) of a block and the (bin, first frame, frame stride) of a lane
struct Step2Geom {
    long long r;
    int c, t0, t1, f, t_lane, t_stride;
    bool nyq;
};

__device__ __forceinline__ Step2Geom step2_geom(const Step2Args& a, int lane) {
    Step2Geom g;
    const int nbin = a.F - 1, tiles = nbin / 64;
    int bid = blockIdx.x;
    g.c = bid % a.chunks;
    bid /= a.chunks;
    const int tile = bid % (tiles + 1);
    g.r = bid / (tiles + 1);
    g.t0 = (int)(((long long)a.T * g.c) / a.chunks);
    g.t1 = (int)(((long long)a.T * (g.c + 1)) / a.chunks);
    g.nyq = tile == tiles;
    g.f = g.nyq ? nbin : tile * 64 + lane;
    g.t_lane = g.nyq ? lane : 0;
    g.t_stride = g.nyq ? 64 : 1;
    return g;
}

template <int M>
__device__ __forceinline__ c32 filt_conj(const c32* w, const c32* x) {      // sum_i conj(w_i) x_i
    c32 z = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < M; ++i) z = cfma_conj(w[i], x[i], z);       // two packed fmas per term (pk.h)
    return z;
}

#ifndef DISCO_S2_U_COV
#define DISCO_S2_U_COV 1
#endif
#ifndef DISCO_S2_U_APPLY
#define DISCO_S2_U_APPLY 4
#endif
constexpr int S2_U_COV = DISCO_S2_U_COV;        // frames per barrier in the covariance kernel (register budget)
constexpr int S2_U_APPLY = DISCO_S2_U_APPLY;    // frames per barrier in the apply kernel

// grid = R * (F/64 + 1) * chunks blocks of 64*K threads.
// SKIPLOC: mask_w is the very array step 1 used (oracle masks, or a DNN's mask reused, tango.py:388-389), so the leading
// M x M block of both step-2 covariances equals the step-1 covariances already sitting in the context as partial sums:
// those 10 of 28 entry pairs (M = 4, K = 4) are neither accumulated nor written, the solver takes them from step 1.
template <int M, int K, bool SKIPLOC>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * K) void k_step2_cov_fused(Step2Args a) {
    constexpr int P = M + K - 1, NP = P * (P + 1) / 2, U = S2_U_COV;
    __shared__ c32 zbuf[2][U][K][64];
    const int k = wave_id(), lane = threadIdx.x & 63;
    const Step2Geom gm = step2_geom(a, lane);
    const int T = a.T, F = a.F, f = gm.f;
    const long long g = gm.r * a.K + k;
    const c32* Xg = a.X + (g * T * (long long)F) * M;
    const float* mg = a.mask + g * T * (long long)F;
    c32* zg = a.z_out ? a.z_out + g * T * (long long)F : nullptr;
    c32 wl[M];
#pragma unroll
    for (int i = 0; i < M; ++i) wl[i] = a.w_loc[(g * F + f) * M + i];
    c32 acc_s[NP], acc_n[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc_s[q] = acc_n[q] = make_float2(0.f, 0.f);
    // software pipeline: the next U frames are requested before the current ones are reduced, so every wave
    // keeps 2*U*(8M+4) bytes per lane in flight while it computes
    auto fetch = [&](int tu, c32 (*xx)[M], float* mm) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tu + u * gm.t_stride + gm.t_lane;
            const bool live = t < gm.t1;
            const long long tf = (long long)(live ? t : gm.t1 - 1) * F + f;      // always a valid frame: loads stay unconditional
#pragma unroll
            for (int i = 0; i < M; ++i) xx[u][i] = Xg[tf * M + i];      // raw: consumed (and zeroed if !live) after the barrier
            mm[u] = mg[tf];
        }
    };
    c32 x[U][M], xn[U][M];
    float m[U], mn[U];
    fetch(gm.t0, x, m);
    int buf = 0;
    for (int tu = gm.t0; tu < gm.t1; tu += U * gm.t_stride, buf ^= 1) {
        fetch(tu + U * gm.t_stride, xn, mn);           // frames >= t1 come back as zeros (predicated off)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tu + u * gm.t_stride + gm.t_lane;
            const c32 z = filt_conj<M>(wl, x[u]);
            zbuf[buf][u][k][lane] = z;
            if (t < gm.t1 && zg) zg[(long long)t * F + f] = z;
        }
        __syncthreads();             // one barrier per U frames; zbuf is double buffered, so none is needed after the reads
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = tu + u * gm.t_stride + gm.t_lane < gm.t1;       // frames past the chunk were loaded clamped: weigh them 0
            const float ms = live ? m[u] : 0.f, mc = live ? 1.f - m[u] : 0.f;
            c32 uu[P];
#pragma unroll
            for (int i = 0; i < M; ++i) uu[i] = x[u][i];
#pragma unroll
            for (int jj = 0; jj < K - 1; ++jj) {       // concatenate_signals order; 'local': remote rows carry this node's mask too
                const int j = jj < k ? jj : jj + 1;
                uu[M + jj] = zbuf[buf][u][j][lane];
            }
            cov_accumulate_shared<P, SKIPLOC ? M : 0>(uu, ms * ms, mc * mc, acc_s, acc_n);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < M; ++i) x[u][i] = xn[u][i];
            m[u] = mn[u];
        }
    }
    // entries of the skipped leading block (never accumulated): q = tri_index(i, j) with j < M
    auto skipped = [](int q) {
        if (!SKIPLOC) return false;
        int i = 0, qq = q;
        while (qq >= P - i) {
            qq -= P - i;
            ++i;
        }
        return i + qq < M;
    };
    if (gm.nyq) {                    // lanes of a wave hold partial sums over disjoint frames of the same (node, bin)
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (skipped(q)) continue;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                acc_s[q].x += __shfl_xor(acc_s[q].x, off);
                acc_s[q].y += __shfl_xor(acc_s[q].y, off);
                acc_n[q].x += __shfl_xor(acc_n[q].x, off);
                acc_n[q].y += __shfl_xor(acc_n[q].y, off);
            }
        }
    }
    if (!gm.nyq || lane == 0) {
        float4* o = a.part + (((g * a.chunks + gm.c) * F) + f) * (long long)NP;
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (!skipped(q)) o[q] = make_float4(acc_s[q].x, acc_s[q].y, acc_n[q].x, acc_n[q].y);
    }
}

template <int M, int K>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * K) void k_step2_apply_fused(Step2Args a) {
    constexpr int P = M + K - 1, U = S2_U_APPLY;
    __shared__ c32 zbuf[2][U][K][64];
    const int k = wave_id(), lane = threadIdx.x & 63;
    const Step2Geom gm = step2_geom(a, lane);
    const int T = a.T, F = a.F, f = gm.f;
    const long long g = gm.r * a.K + k;
    const c32* Xg = a.X + (g * T * (long long)F) * M;
    c32* zg = a.z_out ? a.z_out + g * T * (long long)F : nullptr;
    c32* yg = a.yf + g * T * (long long)F;
    c32 wl[M], wg[P];
#pragma unroll
    for (int i = 0; i < M; ++i) wl[i] = a.w_loc[(g * F + f) * M + i];
#pragma unroll
    for (int i = 0; i < P; ++i) wg[i] = a.w_glo[(g * F + f) * P + i];
    int buf = 0;
    for (int tu = gm.t0; tu < gm.t1; tu += U * gm.t_stride, buf ^= 1) {
        c32 x[U][M];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tu + u * gm.t_stride + gm.t_lane;
            const bool live = t < gm.t1;
            const long long tf = (long long)(live ? t : gm.t1 - 1) * F + f;      // always a valid frame: loads stay unconditional
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const c32 v = Xg[tf * M + i];
                x[u][i] = live ? v : make_float2(0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tu + u * gm.t_stride + gm.t_lane;
            const c32 z = filt_conj<M>(wl, x[u]);
            zbuf[buf][u][k][lane] = z;
            if (t < gm.t1 && zg) zg[(long long)t * F + f] = z;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = tu + u * gm.t_stride + gm.t_lane;
            c32 y = filt_conj<M>(wg, x[u]);
#pragma unroll
            for (int jj = 0; jj < K - 1; ++jj) {
                const int j = jj < k ? jj : jj + 1;
                const c32 z = zbuf[buf][u][j][lane];
                y.x = fmaf(wg[M + jj].x, z.x, fmaf(wg[M + jj].y, z.y, y.x));
                y.y = fmaf(wg[M + jj].x, z.y, fmaf(-wg[M + jj].y, z.x, y.y));
            }
            if (t < gm.t1) yg[(long long)t * F + f] = y;
        }
    }
}

}  // namespace disco

// ---- step-2 filter + iSTFT in one pass over X (tango.py:445 + 528) ------------------------------------------------
// yf never touches HBM: wave k of a workgroup streams node k's STFT frames two at a time for ALL bins (lane owns
// f = lane + 64 j), filters them (z exchange through LDS as above), packs the two filtered frames into one complex
// inverse FFT (fft.h), windows, overlap-adds against the half frame it carries in registers and stores hop segments.
// A workgroup owns `pairs` frame pairs = 2*pairs - 1 hop segments of one room; consecutive workgroups overlap by
// one frame.  w_loc lives in LDS (shared by the frames of the room), w_glo in registers.
#include "k_stft.h"

namespace disco {

// frame pairs per workgroup are a launch parameter (`pairs`): 2*pairs - 1 hop segments per workgroup

// ---- synthesis window + overlap-add of one inverse transform (two frames A = tA, B = tA + 1 ride it: time frame A is
// Re v, frame B is -Im v; v = FFT(conj V), n = lane + 64 e) ------------------------------------------------------------
// out[seg H + n] = (frame seg's upper half + frame seg+1's lower half) / (w[n+H]^2 + w[n]^2), each half weighted by the
// window and 1/N.  All three factors depend on the lane only, so they are folded into two weights per sample position once
// per kernel (cA: lower-half contributions, cB: upper-half): an output sample is one multiply and one fma -- the
// per-sample window products, the sum of squares and an IEEE division (~14 instructions) were spent here before.  The last
// segment of a signal has no successor frame and is normalised by w[n+H]^2 alone: corrected on that segment only.
template <int N>
struct OlaWeights {
    static constexpr int EH = FftPlan<N>::E / 2;
    float cA[EH], cB[EH];
    __device__ __forceinline__ void init(const float* __restrict__ win, int lane) {
        const float inv = 1.0f / N;
#pragma unroll
        for (int e = 0; e < EH; ++e) {
            const float wl = win[lane + 64 * e], wh = win[lane + 64 * (e + EH)];
            const float s = wh * wh + wl * wl;
            const float rn = s > 1.17549435e-38f ? 1.0f / s : 1.0f;
            cA[e] = wl * inv * rn;
            cB[e] = wh * inv * rn;
        }
    }
};
template <int N>
__device__ __forceinline__ float ola_last_segment_factor(const float* __restrict__ win, int lane, int e) {
    constexpr int EH = FftPlan<N>::E / 2;
    const float wl = win[lane + 64 * e], wh = win[lane + 64 * (e + EH)];
    const float a = wh * wh, s = a + wl * wl;
    const float rn = s > 1.17549435e-38f ? 1.0f / s : 1.0f;
    const float rl = a > 1.17549435e-38f ? 1.0f / a : 1.0f;
    return rl / rn;
}
// emits hop segments tA - 1 (carry + lower half of A; skipped for a run's first pair, it belongs to the predecessor) and tA
// (upper half of A + lower half of B); carry <- upper half of B
template <int N>
__device__ __forceinline__ void ola_emit_pair(const c32* v, float* carry, const OlaWeights<N>& ow, float* __restrict__ og,
                                              const float* __restrict__ win, int tA, bool first_pair, int T, int L, int lane) {
    constexpr int EH = FftPlan<N>::E / 2, H = N / 2;
#pragma unroll
    for (int which = 0; which < 2; ++which) {
        const int seg = tA - 1 + which;
        const bool emit = (which == 1 || !first_pair) && seg >= 0 && seg < T;       // wave-uniform
        float val[EH];
#pragma unroll
        for (int e = 0; e < EH; ++e)
            val[e] = which == 0 ? fmaf(v[e].x, ow.cA[e], carry[e]) : fmaf(-v[e].y, ow.cA[e], v[e + EH].x * ow.cB[e]);
        if (emit) {
            if (seg + 1 >= T) {
#pragma unroll
                for (int e = 0; e < EH; ++e) val[e] *= ola_last_segment_factor<N>(win, lane, e);
            }
            float* o = og + (long long)seg * H + lane;
            if ((long long)(seg + 1) * H <= L) {
#pragma unroll
                for (int e = 0; e < EH; ++e) store_stream4(&o[64 * e], val[e]);
            } else {
#pragma unroll
                for (int e = 0; e < EH; ++e)
                    if ((long long)seg * H + lane + 64 * e < L) store_stream4(&o[64 * e], val[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < EH; ++e) carry[e] = -v[e + EH].y * ow.cB[e];
}

#ifndef DISCO_AI_PREFETCH
#define DISCO_AI_PREFETCH 1
#endif

template <int N, int M, int K>
struct alignas(16) ApplyIstftShared {
    c32 buf[K][fft_buf_len<N>()];
    c32 zbuf[2][K][K > 1 ? N / 2 + 1 : 1];      // K = 1 (single node): nothing is exchanged, w_glo = w_loc lives in registers
    c32 wl[K][K > 1 ? N / 2 + 1 : 1][M];
};

// 2 waves per SIMD is what the 68 kB of LDS allow; stated so that the allocator keeps the prefetch within 256 registers
template <int N, int M, int K>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * K, (M <= 4 && K <= 4) ? 2 : 1) void k_step2_apply_istft(Step2Args a, float* __restrict__ out,
                                                               const float* __restrict__ win, const c32* __restrict__ tw,
                                                               int L, int blocks_per_room, int pairs) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1, H = N / 2, EH = E / 2, NJ = EH + 1, P = M + K - 1;
    __shared__ ApplyIstftShared<N, M, K> sh;
    const int k = wave_id(), lane = threadIdx.x & 63;
    const long long r = blockIdx.x / blocks_per_room;
    const int s0 = (int)(blockIdx.x % blocks_per_room) * (2 * pairs - 1);       // first hop segment == first frame
    const int T = a.T;
    const long long g = r * K + k;
    const c32* Xg = a.X + (g * T * (long long)F) * M;
    // the room's local filters -> LDS (every wave needs all of them only through z; its own row is read per frame)
    if constexpr (K > 1) {
        const c32* src = a.w_loc + (r * K) * (long long)F * M;
        c32* dst = &sh.wl[0][0][0];
        for (int i = threadIdx.x; i < K * F * M; i += 64 * K) dst[i] = src[i];
    }
    c32 wg[NJ][P];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int f = (j < EH) ? lane + 64 * j : F - 1;
#pragma unroll
        for (int i = 0; i < P; ++i) wg[j][i] = a.w_glo[(g * F + f) * P + i];
    }
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    OlaWeights<N> ow;
    ow.init(win, lane);
    float carry[EH];
#pragma unroll
    for (int e = 0; e < EH; ++e) carry[e] = 0.f;
    float* og = out + g * (long long)L;
    constexpr bool PF = DISCO_AI_PREFETCH && M <= 4 && K <= 4;     // larger shapes have no registers to spare for it
    // The kernel runs 2 waves per SIMD (LDS- and register-bound) and measured 32 % VALU-busy: latency-bound.  The frame
    // pair's spectra are therefore fetched one pair AHEAD, raw and unconditionally (frame index clamped), into registers
    // that are dead during the inverse FFT, and pinned (DISCO_CONSUME) before the pair's output stores are issued.
    c32 xr[PF ? 2 : 1][PF ? NJ : 1][M];
    auto fetch_pair = [&](int tA_) {
        if constexpr (!PF) return;
#pragma unroll
        for (int fr = 0; fr < (PF ? 2 : 0); ++fr) {
            const long long tf0 = (long long)min(tA_ + fr, T - 1) * F;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int f = (j < EH) ? lane + 64 * j : F - 1;
#pragma unroll
                for (int i = 0; i < M; ++i) xr[fr][j][i] = Xg[(tf0 + f) * M + i];
            }
        }
    };
    fetch_pair(s0);
    if constexpr (K > 1) __syncthreads();
    for (int pr = 0; pr < pairs; ++pr) {
        const int tA = s0 + 2 * pr;
        c32 yf[2][NJ];
        // ---- local part of yf and z for both frames
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
            const int t = tA + fr;
            const bool tv = t < T;
            const long long tf0 = (long long)(tv ? t : T - 1) * F;
            (void)tf0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int f = (j < EH) ? lane + 64 * j : F - 1;
                c32 x[M];
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    c32 v;
                    if constexpr (PF) v = xr[PF ? fr : 0][PF ? j : 0][i];
                    else v = Xg[(tf0 + f) * M + i];
                    x[i] = v;                          // frames past the signal: clamped (finite) data, their yf is zeroed below
                }
                if constexpr (K > 1) {
                    c32 wl[M];
#pragma unroll
                    for (int i = 0; i < M; ++i) wl[i] = sh.wl[k][f][i];
                    const c32 z = filt_conj<M>(wl, x);
                    if (j < EH || lane == 0) sh.zbuf[fr][k][f] = z;
                }
                yf[fr][j] = filt_conj<M>(wg[j], x);
            }
        }
        if constexpr (K > 1) __syncthreads();      // K = 1 (single node, w_glo = w_loc): one wave, nothing to exchange
        // ---- remote rows: yf += sum_jj conj(wg[M+jj]) z_j   (concatenate_signals order)
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int f = (j < EH) ? lane + 64 * j : F - 1;
#pragma unroll
                for (int jj = 0; jj < K - 1; ++jj) {
                    const int jn = jj < k ? jj : jj + 1;
                    const c32 z = sh.zbuf[fr][jn][f];
                    const c32 ww = wg[j][M + jj];
                    yf[fr][j] = cfma_conj(ww, z, yf[fr][j]);
                }
            }
        }
        if (tA + 1 >= T) {                           // wave-uniform, last pair of a signal only: frames past the end contribute nothing
#pragma unroll
            for (int fr = 0; fr < 2; ++fr)
                if (tA + fr >= T) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) yf[fr][j] = make_float2(0.f, 0.f);
                }
        }
        // ---- V = A~ + i B~ (Hermitian extensions of the two frames), conjugated for the inverse-by-forward trick
        c32* buf = sh.buf[k];
        c32 v[E];
        irfft_pair_pack<N>(yf[0], yf[1], v, lane);       // cross-lane, no LDS round trip (fft.h)
        fetch_pair(tA + 2);                               // next pair, in flight during the FFT (harmless clamp at the end)
        fft_wave<N>(v, wtw, buf, lane);
        if constexpr (PF) {
#pragma unroll
            for (int fr = 0; fr < 2; ++fr)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < M; ++i) {
                        DISCO_CONSUME(xr[PF ? fr : 0][PF ? j : 0][i].x);
                        DISCO_CONSUME(xr[PF ? fr : 0][PF ? j : 0][i].y);
                    }
        }
        // ---- window, overlap-add: segment (tA-1) = carry + A[lo], segment tA = A[hi] + B[lo], carry <- B[hi]
        ola_emit_pair<N>(v, carry, ow, og, win, tA, pr == 0, T, L, lane);
        // zbuf is rewritten by the next pair only after every wave has passed the next __syncthreads... but a fast wave
        // could reach its z stores of pair pr+1 while a slow one still reads zbuf of pair pr: fence the reuse
        if constexpr (K > 1) __syncthreads();
    }
}

// ---- the final filter + iSTFT of the WIDE shapes in one pass (tango.py:445 + 528; P = M + K - 1 > 8; round 4) ------------------------------
// disco_apply(X, z, w_glo) followed by disco_istft wrote yf (8 F bytes per node-frame) and read it back: 4.3 GB of C5's step, and the iSTFT
// pass ran at half the rate of the filter pass.  The narrow kernel above keeps a node's filters in registers with a lane owning 9 bins; 9 x 15
// complex taps do not fit.  Here a workgroup serves one node and NR = N / 256 consecutive runs of frame pairs with TWO KINDS OF WAVES:
//   WF = N / 128 filter waves: wave w owns the BINS 64 w + lane (k_apply_mq's layout: the P taps of one bin in registers, the node's spectra
//            fetched as contiguous 16-byte granules through a wave-private swizzled LDS tile, the K - 1 remote z rows as contiguous rows) and
//            filters, per step, the frame pair of EVERY run: 2 NR frames -> one of two LDS rings [2 NR][F].  Its loads run two frames ahead
//            and never pause;
//   NR transform waves: wave x owns RUN x (k_step2_apply_istft's layout: lane holds bins lane + 64 j of both frames of its pair, read from
//            the ring the filter waves have just completed), packs the pair into one complex inverse FFT, windows, overlap-adds against the
//            half frame it carries in registers, stores hop segments -- while the filter waves fill the other ring.
// ONE barrier per step: the filter waves arrive with ring pr & 1 written, the transform waves with ring (pr - 1) & 1 read.
// (The first version gave every wave both roles in turn -- 16 frames filtered, barrier, one transform per wave, barrier: its loads stood
// still during the transforms, 0.8 of its 4.3 ms per C5 launch, and the second frame's look-ahead had to be re-issued behind them for lack of
// registers: profiles/r04_v_wide_parts.txt.  This form: 4.36 against 4.62 + 1.07 ms for disco_apply + disco_istft on the same box, 21.7 GB
// of traffic at 5 TB/s: the read rate of the filter pass itself.)
// The Nyquist bin (the 513th of 1024 / 2 + 1) has no lane in the WF full tiles: filter wave w handles it for frame w of the step, one value
// per lane loaded at the top of the step, lane 0 doing the arithmetic; its taps sit in LDS.
// The arithmetic of a bin is k_apply_mq's instruction for instruction, the transform and the overlap-add are k_step2_apply_istft's.
// grid: (chunk, room, node) items, node fastest, dealt XCD-aware (the nodes of a room read the same z rows); a workgroup covers
// NR * (2 * pairs - 1) hop segments, consecutive runs overlap by one frame.
struct ApplyIstftWideArgs {
    const c32* X;        // [R][K][T][F][M]
    const c32* Z;        // [R][K][T][F]
    const c32* w;        // [R][K][F][P]
    float* out;          // [R][K][L]
    c32* yf;             // [R][K][T][F] or NULL (only when the caller wants the filtered spectra)
    int T, L, pairs, chunks;
    long long R;
    // a node shard (disco_set_node_shard / disco_set_z_blocks): X, w, out, yf hold the Kl nodes [k0, k0 + Kl) of every room, Z the z of ALL K
    // nodes in planes [K / zblk][R][zblk] (z_plane, common.h); the whole room: Kl = K, k0 = 0, zblk = K
    int Kl, k0, zblk;
};
template <int N, int M, int KR>
struct alignas(16) ApplyIstftWideShared {
    static constexpr int WF = N / 128, NR = N / 256, F = N / 2 + 1;
    static constexpr int NT1 = WaveTw<N>::Q1 * (FftPlan<N>::R1 - 1), NT2 = WaveTw<N>::Q2 * (FftPlan<N>::R2 - 1), EH = FftPlan<N>::E / 2;
    static constexpr int NL = M / 2 + KR;               // loads of one frame's Nyquist bin: M / 2 granules of X + KR remote z's
    float4 tile[WF][64 * (M / 2)];                      // a filter wave's granules of one frame, swizzled
    c32 buf[NR][fft_buf_len<N>()];                      // a transform wave's FFT scratch
    c32 ring[2][2 * NR][F + 1];
    c32 twl[NT1 + NT2][64];                             // WaveTw<N> and OlaWeights<N>, lane-minor: the same for every transform wave, loaded per pass
    float olw[2 * EH][64];
    float4 nyq[WF][NL];
    c32 wn[M + KR + 1];
};
#ifndef DISCO_WIDE_EXP
#define DISCO_WIDE_EXP 0                                // TIMING-ONLY builds: bit 0 no transform / overlap-add, 1 no filter arithmetic, 2 no loads (garbage results)
#endif
template <int N, int M, int KR>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * (N / 128 + N / 256), 1) void k_apply_istft_wide(ApplyIstftWideArgs a, const float* __restrict__ win,
                                                                                                     const c32* __restrict__ tw) {
    using Sh = ApplyIstftWideShared<N, M, KR>;
    constexpr int WF = Sh::WF, NR = Sh::NR, E = FftPlan<N>::E, F = N / 2 + 1, EH = E / 2, NJ = EH + 1, P = M + KR, K = KR + 1;
    constexpr int MH = M / 2, BPR = 16 / MH, NL = Sh::NL;
    static_assert(M == 4 || M == 8, "whole 256-byte bank rows per group of bins");
    static_assert(2 * NR == WF && NL <= 64, "filter wave w takes the Nyquist bin of frame w of a step");
    __shared__ Sh sh;
    const int wv = wave_id(), lane = threadIdx.x & 63;
    const int T = a.T;
    const long long n_items = a.R * a.Kl * (long long)a.chunks;
    long long item = (long long)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    if (item >= n_items) return;
    const int kl = (int)(item % a.Kl);
    item /= a.Kl;
    const long long r = item % a.R;
    const int chunk = (int)(item / a.R);
    const int k = a.k0 + kl;                                                 // the node's place in its room (which z rows are remote)
    const long long g = r * a.Kl + kl;
    const int run_len = 2 * a.pairs - 1;
    const int s_base = chunk * NR * run_len;                                 // run u starts at frame (= hop segment) s_base + u * run_len
    const long long TF = (long long)T * F;
    auto frame_of = [&](int pr, int u, int fr) { return s_base + u * run_len + 2 * pr + fr; };
    // tables every wave reads: the Nyquist bin's taps (conjugated once), the transform waves' twiddles and overlap-add weights
    if (threadIdx.x < P) {
        const c32 t_ = a.w[(g * F + (F - 1)) * (long long)P + threadIdx.x];
        sh.wn[threadIdx.x] = make_float2(t_.x, -1.f * t_.y);
    }
    if (wv == WF) {
        WaveTw<N> w0;
        w0.init(tw, lane);
        OlaWeights<N> o0;
        o0.init(win, lane);
#pragma unroll
        for (int i = 0; i < Sh::NT1; ++i) sh.twl[i][lane] = w0.t1[i];
#pragma unroll
        for (int i = 0; i < Sh::NT2; ++i) sh.twl[Sh::NT1 + i][lane] = w0.t2[i];
#pragma unroll
        for (int e = 0; e < EH; ++e) {
            sh.olw[e][lane] = o0.cA[e];
            sh.olw[EH + e][lane] = o0.cB[e];
        }
    }
    if (wv < WF) {
        // ================= filter wave: bin f of tile wv, its taps (conjugated once, as k_apply_mq with conj_w = 1)
        const int f = wv * 64 + lane;
        const c32* wf = a.w + (g * F + f) * (long long)P;
        c32 wl[M], wr[KR];
#pragma unroll
        for (int i = 0; i < M; ++i) wl[i] = make_float2(wf[i].x, -1.f * wf[i].y);
        const c32* zrow[KR];                                                 // (wave-uniform) remote rows in concatenate_signals order
#pragma unroll
        for (int jj = 0; jj < KR; ++jj) {
            wr[jj] = make_float2(wf[M + jj].x, -1.f * wf[M + jj].y);
            zrow[jj] = a.Z + z_plane(r, jj < k ? jj : jj + 1, K, a.R, a.zblk) * TF;
        }
        // granule r * 64 + lane of the tile's frame = (bin b, granule lane % MH); LDS position of that granule (k_apply_mq)
        int lgo[MH], lpos[MH];
#pragma unroll
        for (int q_ = 0; q_ < MH; ++q_) {
            const int gi = q_ * 64 + lane, b = gi / MH, pp = gi % MH;
            lgo[q_] = (wv * 64 + b) * MH + pp;
            lpos[q_] = b * MH + (pp ^ ((b / BPR) % MH));
        }
        const int swz = (lane / BPR) % MH;
        float4* sx = sh.tile[wv];
        const float4* Xq = reinterpret_cast<const float4*>(a.X + g * TF * M);
        c32* yg = a.yf ? a.yf + g * TF : nullptr;
        float q[2][MH][4];                              // (scalars on purpose, see k_apply_mq)
        c32 zq[2][KR];
        auto fetch = [&](int pr, int u, int fr) {
            if (DISCO_WIDE_EXP & 4) return;
            const long long tF = (long long)min(frame_of(pr, u, fr), T - 1) * F;   // frames past the signal: clamped (finite) data, zeroed at the ring
#pragma unroll
            for (int q_ = 0; q_ < MH; ++q_) {
                const float4 v = Xq[tF * MH + lgo[q_]];
                q[fr][q_][0] = v.x;
                q[fr][q_][1] = v.y;
                q[fr][q_][2] = v.z;
                q[fr][q_][3] = v.w;
            }
#pragma unroll
            for (int jj = 0; jj < KR; ++jj) zq[fr][jj] = zrow[jj][tF + f];
        };
        // Nyquist bin of frame wv of a step (run wv / 2, frame wv % 2): load i sits in lane i (a granule of X or one remote z)
        const bool n_ld = lane < NL, n_isx = lane < MH;
        const int n_jj = max(lane - MH, 0);
        const float* n_src = n_isx ? reinterpret_cast<const float*>(Xq + (long long)(F - 1) * MH + lane)
                                   : reinterpret_cast<const float*>(a.Z + z_plane(r, n_jj < k ? n_jj : n_jj + 1, K, a.R, a.zblk) * TF + (F - 1));
        const long long n_stride = n_isx ? (long long)F * M * 2 : (long long)F * 2;        // floats per frame
        fetch(0, 0, 0);
        fetch(0, 0, 1);
        __syncthreads();                                // wn (and the transform waves' tables)
        for (int pr = 0; pr < a.pairs; ++pr) {
            c32(*ring)[F + 1] = sh.ring[pr & 1];
            const int tn = frame_of(pr, wv / 2, wv & 1);
            float nv[4] = {0.f, 0.f, 0.f, 0.f};
            if (n_ld && !(DISCO_WIDE_EXP & 4)) {
                const float* p_ = n_src + (long long)min(tn, T - 1) * n_stride;
                if (n_isx) {
                    const float4 v = *reinterpret_cast<const float4*>(p_);
                    nv[0] = v.x;
                    nv[1] = v.y;
                    nv[2] = v.z;
                    nv[3] = v.w;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(p_);
                    nv[0] = v.x;
                    nv[1] = v.y;
                }
            }
            for (int u = 0; u < NR; ++u) {
                const bool wrap = u + 1 == NR;
                const int pn = wrap ? min(pr + 1, a.pairs - 1) : pr, un = wrap ? 0 : u + 1;     // (the last pair again at the very end)
#pragma unroll
                for (int fr = 0; fr < 2; ++fr) {
                    DISCO_LDS_WAR();                    // the previous frame's reads of the tile are over
#pragma unroll
                    for (int q_ = 0; q_ < MH; ++q_) sx[lpos[q_]] = make_float4(q[fr][q_][0], q[fr][q_][1], q[fr][q_][2], q[fr][q_][3]);
                    c32 z[KR];
#pragma unroll
                    for (int jj = 0; jj < KR; ++jj) z[jj] = zq[fr][jj];
                    DISCO_LDS_RAW();
                    fetch(pn, un, fr);                  // the same frame slot of the next pair is on its way while this one is filtered
                    c32 x[M];
#pragma unroll
                    for (int pp = 0; pp < MH; ++pp) {
                        const float4 v = sx[lane * MH + (pp ^ swz)];
                        x[2 * pp] = make_float2(v.x, v.y);
                        x[2 * pp + 1] = make_float2(v.z, v.w);
                    }
                    DISCO_LDS_RAW();
                    float ar = 0.f, ai = 0.f;
                    if (!(DISCO_WIDE_EXP & 2)) {
#pragma unroll
                        for (int i = 0; i < M; ++i) {
                            ar = fmaf(wl[i].x, x[i].x, fmaf(-wl[i].y, x[i].y, ar));
                            ai = fmaf(wl[i].x, x[i].y, fmaf(wl[i].y, x[i].x, ai));
                        }
#pragma unroll
                        for (int jj = 0; jj < KR; ++jj) {
                            ar = fmaf(wr[jj].x, z[jj].x, fmaf(-wr[jj].y, z[jj].y, ar));
                            ai = fmaf(wr[jj].x, z[jj].y, fmaf(wr[jj].y, z[jj].x, ai));
                        }
                    } else {
                        ar = x[0].x + z[0].x;
                        ai = x[M - 1].y + z[KR - 1].y;
                    }
                    const int t = frame_of(pr, u, fr);
                    ring[2 * u + fr][f] = t < T ? make_float2(ar, ai) : make_float2(0.f, 0.f);
                    if (yg && t < T) yg[(long long)t * F + f] = make_float2(ar, ai);
                }
            }
            if (n_ld) sh.nyq[wv][lane] = make_float4(nv[0], nv[1], nv[2], nv[3]);
            DISCO_LDS_RAW();
            if (lane == 0) {
                const float4* nb = sh.nyq[wv];
                float ar = 0.f, ai = 0.f;
#pragma unroll
                for (int pp = 0; pp < MH; ++pp) {
                    const float4 v = nb[pp];
                    const c32 w0_ = sh.wn[2 * pp], w1_ = sh.wn[2 * pp + 1];
                    ar = fmaf(w0_.x, v.x, fmaf(-w0_.y, v.y, ar));
                    ai = fmaf(w0_.x, v.y, fmaf(w0_.y, v.x, ai));
                    ar = fmaf(w1_.x, v.z, fmaf(-w1_.y, v.w, ar));
                    ai = fmaf(w1_.x, v.w, fmaf(w1_.y, v.z, ai));
                }
#pragma unroll
                for (int jj = 0; jj < KR; ++jj) {
                    const c32 w_ = sh.wn[M + jj];
                    const float4 v = nb[MH + jj];
                    ar = fmaf(w_.x, v.x, fmaf(-w_.y, v.y, ar));
                    ai = fmaf(w_.x, v.y, fmaf(w_.y, v.x, ai));
                }
                ring[wv][F - 1] = tn < T ? make_float2(ar, ai) : make_float2(0.f, 0.f);
                if (yg && tn < T) yg[(long long)tn * F + (F - 1)] = make_float2(ar, ai);
            }
            __syncthreads();                            // ring pr & 1 is complete; the transform waves have read ring (pr - 1) & 1
        }
    } else {
        // ================= transform wave: run x
        const int x_ = wv - WF;
        float carry[EH];
#pragma unroll
        for (int e = 0; e < EH; ++e) carry[e] = 0.f;
        float* og = a.out + g * (long long)a.L;
        const int s0 = s_base + x_ * run_len;
        __syncthreads();                                // (tables)
        for (int pr = 0; pr < a.pairs; ++pr) {
            __syncthreads();                            // the filter waves have completed ring pr & 1
            const c32(*ring)[F + 1] = sh.ring[pr & 1];
            const int tA = s0 + 2 * pr;
            if (tA - 1 < T && !(DISCO_WIDE_EXP & 1)) {  // (wave-uniform) else: a run past the signal's end, nothing left to emit
                c32 yf[2][NJ];
#pragma unroll
                for (int fr = 0; fr < 2; ++fr) {
#pragma unroll
                    for (int j = 0; j < EH; ++j) yf[fr][j] = ring[2 * x_ + fr][lane + 64 * j];
                    yf[fr][EH] = ring[2 * x_ + fr][F - 1];
                }
                c32 v[E];
                irfft_pair_pack<N>(yf[0], yf[1], v, lane);
                {                                       // fft_wave<N> with each pass' twiddles fetched from LDS when the pass needs them
                    using Pl = FftPlan<N>;
                    c32* buf = sh.buf[x_];
                    DISCO_LDS_WAR();
                    fft_pass<N, Pl::R0, 1, true, false>(v, nullptr, buf, lane);
                    {
                        c32 t1[Sh::NT1];
#pragma unroll
                        for (int i = 0; i < Sh::NT1; ++i) t1[i] = sh.twl[i][lane];
                        fft_pass<N, Pl::R1, Pl::R0, false, false>(v, t1, buf, lane);
                    }
                    {
                        c32 t2[Sh::NT2];
#pragma unroll
                        for (int i = 0; i < Sh::NT2; ++i) t2[i] = sh.twl[Sh::NT1 + i][lane];
                        fft_pass<N, Pl::R2, Pl::R0 * Pl::R1, false, true>(v, t2, buf, lane);
                    }
                }
                OlaWeights<N> ow;
#pragma unroll
                for (int e = 0; e < EH; ++e) {
                    ow.cA[e] = sh.olw[e][lane];
                    ow.cB[e] = sh.olw[EH + e][lane];
                }
                ola_emit_pair<N>(v, carry, ow, og, win, tA, pr == 0, T, a.L, lane);
            }
        }
    }
}

// ---- single node: STFT -> filter -> iSTFT in one pass over the SAMPLES (config C2; get_z_signals.py:274-315 + tango.py:528) ------
// With one node there is no exchange and step 2 repeats step 1, so after the statistics pass (k_stft_cov<.., STORE = false>)
// and the solve, the output is iSTFT(w^H STFT(y)).  Reading the spectra back would move 8 M F bytes per node-frame; recomputing
// them from the 4 M H bytes of samples they came from moves a quarter of that, and nothing but the time signal is written: the
// pass is pure wave-local arithmetic (M/2 forward transforms per frame, one inverse per frame pair, no workgroup barrier).
// A wave owns `pairs` frame pairs of one node (2 pairs - 1 hop segments; consecutive waves overlap by one frame, as in
// k_step2_apply_istft); the half-window shared by consecutive frames is recycled in registers as in k_stft.
// DISCO_SAI_EXP (default 0): TIMING-ONLY builds of k_stft_apply_istft -- bit 0 no forward transforms (the filter takes the windowed samples),
// 1 no inverse transform, 2 no output stores, 3 no sample loads; results are garbage (profiles/r05_h_stft_parts.txt)
#ifndef DISCO_SAI_EXP
#define DISCO_SAI_EXP 0
#endif
// DISCO_SAI_TAPS_LDS (default 1): the node's filter -- (N/2 + 1) x M taps, 40 registers per lane for a 512-point, 4-mic node -- sits in a
// wave-private LDS block [pair][bin] (one conflict-free ds_read_b128 per channel pair and bin) instead of registers: 198 -> under 168 registers,
// THREE waves per SIMD instead of two.  The pass is bound by the latency of its loads and of its transforms at two waves per SIMD
// (profiles/r05_h_stft_parts.txt), so the third wave is what pays.
#ifndef DISCO_SAI_TAPS_LDS
#define DISCO_SAI_TAPS_LDS 1
#endif
template <int N, int CHP>
struct alignas(16) StftApplyShared {
    c32 buf[STFT_WAVES][fft_buf_len<N>()];
    float4 taps[DISCO_SAI_TAPS_LDS ? STFT_WAVES : 1][DISCO_SAI_TAPS_LDS ? CHP : 1][DISCO_SAI_TAPS_LDS ? N / 2 + 1 : 1];
};
template <int N, int M>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64 * STFT_WAVES, M <= 4 ? (DISCO_SAI_TAPS_LDS ? 3 : 2) : 1) void k_stft_apply_istft(const float* __restrict__ x, const c32* __restrict__ wf,
                                                                                    float* __restrict__ out, const float* __restrict__ win,
                                                                                    const c32* __restrict__ tw, int L, int T, int pad_mode,
                                                                                    int runs_per_node, int pairs, long long n_witems) {
    constexpr int E = FftPlan<N>::E, F = N / 2 + 1, H = N / 2, EH = E / 2, NJ = EH + 1, CHP = (M + 1) / 2;
    constexpr bool TL = DISCO_SAI_TAPS_LDS != 0;
    __shared__ StftApplyShared<N, CHP> sh;
    const int wave = wave_id(), lane = threadIdx.x & 63;
    const long long item = (long long)blockIdx.x * STFT_WAVES + wave;
    if (item >= n_witems) return;                  // no block-level synchronisation anywhere below
    const long long g = item / runs_per_node;
    const int s0 = (int)(item % runs_per_node) * (2 * pairs - 1);             // first hop segment == first frame
    c32* buf = sh.buf[wave];
    WaveTw<N> wtw;
    wtw.init(tw, lane);
    float w[E];
    load_window<N>(w, win, lane);
    // the node's filter at this lane's bins (f = lane + 64 j; lane 0: Nyquist too), at half weight: the forward transforms below run on
    // the full-weight window (it is the synthesis window too), so the untangled spectra are 2 X (fft.h: rfft_pair_untangle leaves the
    // division by two to its caller); exact
    c32 wg[TL ? 1 : NJ][TL ? 1 : M];
    if constexpr (TL) {
        for (int f = lane; f < F; f += 64)
#pragma unroll
            for (int p = 0; p < CHP; ++p) {
                const c32 a = wf[(g * F + f) * M + 2 * p], b = (2 * p + 1 < M) ? wf[(g * F + f) * M + 2 * p + 1] : make_float2(0.f, 0.f);
                sh.taps[wave][p][f] = make_float4(0.5f * a.x, 0.5f * a.y, 0.5f * b.x, 0.5f * b.y);
            }
        DISCO_LDS_RAW();                             // wave-private block: the wave's own writes are in place
    } else {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int f = (j < EH) ? lane + 64 * j : F - 1;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const c32 wv = wf[(g * F + f) * M + i];
                wg[j][i] = make_float2(0.5f * wv.x, 0.5f * wv.y);
            }
        }
    }
    const float* xa[CHP];
    const float* xb[CHP];
#pragma unroll
    for (int p = 0; p < CHP; ++p) {
        xa[p] = x + (g * M + 2 * p) * (long long)L;
        xb[p] = (2 * p + 1 < M) ? x + (g * M + 2 * p + 1) * (long long)L : xa[p];
    }
    c32 raw[CHP][E];
#pragma unroll
    for (int p = 0; p < CHP; ++p) load_frame_slots<N, 0, E>(raw[p], xa[p], xb[p], min(s0, T - 1), L, pad_mode, lane);
    OlaWeights<N> ow;
    ow.init(win, lane);
    float carry[EH];
#pragma unroll
    for (int e = 0; e < EH; ++e) carry[e] = 0.f;
    float* og = out + g * (long long)L;
    for (int pr = 0; pr < pairs; ++pr) {
        const int tA = s0 + 2 * pr;
        c32 yf[2][NJ];
#pragma unroll
        for (int fr = 0; fr < 2; ++fr) {
            const int t = tA + fr;
            c32 nxt[CHP][EH];                        // the next frame's new half-window, in flight under this frame's transforms
#pragma unroll
            for (int p = 0; p < CHP; ++p) {
                if (DISCO_SAI_EXP & 8) {
#pragma unroll
                    for (int e = 0; e < EH; ++e) nxt[p][e] = make_float2(0.25f * e + lane, 1.f);
                } else {
                    load_frame_slots<N, EH, E>(nxt[p], xa[p], xb[p], min(t + 1, T - 1), L, pad_mode, lane);
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) yf[fr][j] = make_float2(0.f, 0.f);
#pragma unroll
            for (int p = 0; p < CHP; ++p) {
                c32 v[E];
                apply_window<N>(v, raw[p], w, 2 * p + 1 < M);
                if (!(DISCO_SAI_EXP & 1)) fft_wave<N>(v, wtw, buf, lane);
                rfft_pair_untangle<N>(v, buf, lane, [&](int j, int f, c32 a, c32 b) {
                    // yf += conj(w_2p) X_2p + conj(w_2p+1) X_2p+1
                    c32 wa, wb;
                    if constexpr (TL) {
                        const float4 t4 = sh.taps[wave][p][f];
                        wa = make_float2(t4.x, t4.y);
                        wb = make_float2(t4.z, t4.w);
                    } else {
                        wa = wg[j][2 * p];
                        wb = wg[j][2 * p + 1 < M ? 2 * p + 1 : 2 * p];
                    }
                    c32 acc = cfma_conj(wa, a, yf[fr][j]);
                    if (2 * p + 1 < M) acc = cfma_conj(wb, b, acc);
                    yf[fr][j] = acc;
                });
            }
            if (t >= T) {                            // frames past the signal contribute nothing
#pragma unroll
                for (int j = 0; j < NJ; ++j) yf[fr][j] = make_float2(0.f, 0.f);
            }
#pragma unroll
            for (int p = 0; p < CHP; ++p)
#pragma unroll
                for (int e = 0; e < EH; ++e) {
                    DISCO_CONSUME(nxt[p][e].x);
                    DISCO_CONSUME(nxt[p][e].y);
                    raw[p][e] = raw[p][e + EH];
                    raw[p][e + EH] = nxt[p][e];
                }
        }
        // ---- V = A~ + i B~ (Hermitian extensions of the two filtered frames), conjugated for the inverse-by-forward trick
        c32 v[E];
        irfft_pair_pack<N>(yf[0], yf[1], v, lane);       // cross-lane, no LDS round trip (fft.h)
        if (!(DISCO_SAI_EXP & 2)) fft_wave<N>(v, wtw, buf, lane);
        // ---- window, overlap-add: segment (tA-1) = carry + A[lo], segment tA = A[hi] + B[lo], carry <- B[hi]
        if (!(DISCO_SAI_EXP & 4)) ola_emit_pair<N>(v, carry, ow, og, win, tA, pr == 0, T, L, lane);
        else if (v[0].x == 123456.f) og[lane] = v[1].y;     // (keeps the arithmetic alive)
    }
}

}  // namespace disco
