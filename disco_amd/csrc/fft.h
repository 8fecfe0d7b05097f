// Wave-level complex FFT for gfx950: one 64-lane wavefront transforms N = 64*E points (E per lane)
// with a Stockham auto-sort radix schedule through a wave-private LDS buffer.
//   N = 512  : E = 8,  radices 8 x 8 x 8
//   N = 1024 : E = 16, radices 16 x 16 x 4
// Real signals ride two-for-one: two real channels are packed as re/im of one complex transform and
// separated afterwards with the Hermitian symmetry (stft_pair / istft_pair below).
//
// Element ownership: before the first pass lane l holds x[l + 64 e] in slot e; after the last pass it holds
// X[l + 64 e] in slot e (natural order), so global loads/stores of consecutive lanes are contiguous.
#pragma once
#include "common.h"

namespace disco {

// LDS exchange fences.  The exchange buffers are WAVE-PRIVATE, so no s_barrier is needed: a wave's DS
// instructions execute in issue order, and all that has to be prevented is (a) the compiler moving a read
// above the writes it depends on through another lane, (b) a read issuing before this wave's own writes have
// been accepted.  RAW = `s_waitcnt lgkmcnt(0)` (0xC07F: vmcnt/expcnt untouched, so global prefetches stay in
// flight) + a scheduling barrier; WAR = scheduling barrier only.
#define DISCO_LDS_RAW()                       \
    do {                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);   \
        __builtin_amdgcn_wave_barrier();      \
    } while (0)
#define DISCO_LDS_WAR() __builtin_amdgcn_wave_barrier()

template <int N>
struct FftPlan;
template <>
struct FftPlan<512> {
    static constexpr int E = 8, PADSH = 3, NPASS = 3;
    static constexpr int R0 = 8, R1 = 8, R2 = 8;
};
template <>
struct FftPlan<1024> {
    static constexpr int E = 16, PADSH = 4, NPASS = 3;
    static constexpr int R0 = 16, R1 = 16, R2 = 4;
};

template <int N>
__device__ __forceinline__ int fft_pad(int a) { return a + (a >> FftPlan<N>::PADSH); }
template <int N>
constexpr int fft_buf_len() { return N + (N >> FftPlan<N>::PADSH); }

// ---- small DFTs, forward sign exp(-2 pi i / R), natural-order output, in place ------------------------
__device__ __forceinline__ void dft4(c32& u0, c32& u1, c32& u2, c32& u3) {
    c32 a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3), a3 = cmul_mi(csub(u1, u3));
    u0 = cadd(a0, a2);
    u2 = csub(a0, a2);
    u1 = cadd(a1, a3);
    u3 = csub(a1, a3);
}

__device__ __forceinline__ void dft8(c32* u) {
    const float h = 0.70710678118654752440f;
    c32 e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6];
    c32 o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    // W8^k * O[k]
    c32 t1 = make_float2(h * (o1.x + o1.y), h * (o1.y - o1.x));     // (1 - i)/sqrt2
    c32 t2 = cmul_mi(o2);                                           // -i
    c32 t3 = make_float2(h * (o3.y - o3.x), -h * (o3.x + o3.y));    // (-1 - i)/sqrt2
    u[0] = cadd(e0, o0);
    u[4] = csub(e0, o0);
    u[1] = cadd(e1, t1);
    u[5] = csub(e1, t1);
    u[2] = cadd(e2, t2);
    u[6] = csub(e2, t2);
    u[3] = cadd(e3, t3);
    u[7] = csub(e3, t3);
}

__device__ __forceinline__ void dft16(c32* u) {
    c32 e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = u[2 * i];
        o[i] = u[2 * i + 1];
    }
    dft8(e);
    dft8(o);
    // W16^k = exp(-2 pi i k / 16), k = 0..7
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    const c32 w[8] = {make_float2(1.f, 0.f),  make_float2(c1, -s1), make_float2(h, -h),  make_float2(s1, -c1),
                      make_float2(0.f, -1.f), make_float2(-s1, -c1), make_float2(-h, -h), make_float2(-c1, -s1)};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        c32 t = cmul(o[k], w[k]);
        u[k] = cadd(e[k], t);
        u[k + 8] = csub(e[k], t);
    }
}

template <int R>
__device__ __forceinline__ void dftR(c32* u) {
    if constexpr (R == 4) dft4(u[0], u[1], u[2], u[3]);
    else if constexpr (R == 8) dft8(u);
    else dft16(u);
}

// Per-lane twiddle factors of passes 2 and 3, gathered once per kernel from the N-entry table
// tw[j] = exp(-2 pi i j / N): they depend on the lane only, so they live in registers for every transform
// the wave performs (saves 14 LDS reads per 512-point transform and the LDS table itself).
template <int N>
struct WaveTw {
    using Pl = FftPlan<N>;
    static constexpr int E = Pl::E;
    static constexpr int Q1 = E / Pl::R1, Q2 = E / Pl::R2;
    c32 t1[Q1 * (Pl::R1 - 1)];
    c32 t2[Q2 * (Pl::R2 - 1)];
    __device__ __forceinline__ void init(const c32* __restrict__ tw, int lane) {
        constexpr int P1 = Pl::R0, P2 = Pl::R0 * Pl::R1;
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            const int k = (lane + 64 * q) & (P1 - 1);
#pragma unroll
            for (int r = 1; r < Pl::R1; ++r) t1[q * (Pl::R1 - 1) + r - 1] = tw[k * r * (N / (P1 * Pl::R1))];
        }
#pragma unroll
        for (int q = 0; q < Q2; ++q) {
            const int k = (lane + 64 * q) & (P2 - 1);
#pragma unroll
            for (int r = 1; r < Pl::R2; ++r) t2[q * (Pl::R2 - 1) + r - 1] = tw[k * r * (N / (P2 * Pl::R2))];
        }
    }
};

// One Stockham pass of radix R with P_ = product of the previous radices.
//   butterfly i in [0, N/R): k = i mod P_; u[r] = x[i + r N/R] * W_{P_ R}^{k r}; U = DFT_R(u); y[(i-k) R + k + r P_] = U[r]
// A lane owns the E/R butterflies i = lane + 64 q; v[q R + r] carries u[r] / U[r].
template <int N, int R, int P_, bool FIRST, bool LAST>
__device__ __forceinline__ void fft_pass(c32* v, const c32* tw, c32* buf, int lane) {
    constexpr int E = FftPlan<N>::E, S = N / R, Q = E / R;
    if constexpr (!FIRST) {
        DISCO_LDS_RAW();
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) v[q * R + r] = buf[fft_pad<N>(lane + 64 * q + r * S)];
        DISCO_LDS_WAR();
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = lane + 64 * q;
        const int k = i & (P_ - 1);
        if constexpr (P_ > 1) {                    // tw = this pass' register twiddles W_{P_ R}^{k r}, r = 1..R-1
#pragma unroll
            for (int r = 1; r < R; ++r) v[q * R + r] = cmul(v[q * R + r], tw[q * (R - 1) + r - 1]);
        }
        dftR<R>(v + q * R);
        if constexpr (!LAST) {
            const int o = (i - k) * R + k;
#pragma unroll
            for (int r = 0; r < R; ++r) buf[fft_pad<N>(o + r * P_)] = v[q * R + r];
        }
    }
    if constexpr (LAST && (Q > 1)) {
        // slot order: X[lane + 64 e] with e = q + r Q currently sits in v[q R + r]
        c32 tmp[E];
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int r = 0; r < R; ++r) tmp[q + r * Q] = v[q * R + r];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = tmp[e];
    }
}

// Forward complex FFT of the wave's N points.  In: v[e] = x[lane + 64 e].  Out: v[e] = X[lane + 64 e].
// `buf` = wave-private LDS of fft_buf_len<N>() c32.
template <int N>
__device__ __forceinline__ void fft_wave(c32* v, const WaveTw<N>& tw, c32* buf, int lane) {
    using Pl = FftPlan<N>;
    DISCO_LDS_WAR();        // the previous user of `buf` (an earlier item's untangle reads) is done in every lane
    fft_pass<N, Pl::R0, 1, true, false>(v, nullptr, buf, lane);
    fft_pass<N, Pl::R1, Pl::R0, false, false>(v, tw.t1, buf, lane);
    fft_pass<N, Pl::R2, Pl::R0 * Pl::R1, false, true>(v, tw.t2, buf, lane);
}

// Spectra of two real sequences a, b from Z = FFT(a + i b):  A[f] = (Z[f] + conj Z[N-f]) / 2,
// B[f] = (Z[f] - conj Z[N-f]) / (2i).  Lane receives f = lane + 64 j (j < E/2) through emit(j, f, A, B);
// lane 0 additionally receives the Nyquist bin f = N/2.
template <int N, class Emit>
__device__ __forceinline__ void rfft_pair_untangle(c32* v, c32* buf, int lane, Emit emit) {
    constexpr int E = FftPlan<N>::E;
    DISCO_LDS_WAR();
#pragma unroll
    for (int e = 0; e < E; ++e) buf[fft_pad<N>(lane + 64 * e)] = v[e];
    DISCO_LDS_RAW();
#pragma unroll
    for (int j = 0; j <= E / 2; ++j) {
        const int f = lane + 64 * j;
        if (j < E / 2 || lane == 0) {
            const c32 z = buf[fft_pad<N>(f)];
            const c32 zc = buf[fft_pad<N>((N - f) & (N - 1))];
            const c32 A = make_float2(0.5f * (z.x + zc.x), 0.5f * (z.y - zc.y));
            const c32 B = make_float2(0.5f * (z.y + zc.y), -0.5f * (z.x - zc.x));
            emit(j, f, A, B);
        }
    }
}

}  // namespace disco
