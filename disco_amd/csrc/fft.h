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
#include "pk.h"

namespace disco {

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
// Written on the packed primitives of pk.h: every +-i rotation rides the add's operand selectors, the 1/sqrt2 rotations of
// the radix-8 step are one packed add (1 -+ i) + one packed fma against the partner each.  dft4: 8 instructions,
// dft8: 26, no moves, no sign flips.
__device__ __forceinline__ void dft4(c32& u0, c32& u1, c32& u2, c32& u3) {
    const c32 a0 = cadd(u0, u2), a1 = csub(u0, u2), a2 = cadd(u1, u3), d = csub(u1, u3);
    u0 = cadd(a0, a2);
    u2 = csub(a0, a2);
    u1 = cadd_mi(a1, d);        // a1 + (-i) d
    u3 = cadd_pi(a1, d);        // a1 - (-i) d
}

__device__ __forceinline__ void dft8(c32* u) {
    const c32 hh = make_float2(0.70710678118654752440f, 0.70710678118654752440f);
    c32 e0 = u[0], e1 = u[2], e2 = u[4], e3 = u[6];
    c32 o0 = u[1], o1 = u[3], o2 = u[5], o3 = u[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    // W8^k O[k]:  W8 = (1 - i)/sqrt2,  W8^2 = -i,  W8^3 = -(1 + i)/sqrt2
    const c32 s1 = cmul_1mi(o1), s3 = cmul_1pi(o3);
    u[0] = cadd(e0, o0);
    u[4] = csub(e0, o0);
    u[1] = cfma_scale(s1, hh, e1);
    u[5] = cfms_scale(s1, hh, e1);
    u[2] = cadd_mi(e2, o2);
    u[6] = cadd_pi(e2, o2);
    u[3] = cfms_scale(s3, hh, e3);
    u[7] = cfma_scale(s3, hh, e3);
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
    // W16^k = exp(-2 pi i k / 16): k = 0, 4 are free, k = 2, 6 the 1/sqrt2 rotations, the odd k full products with constants
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    const c32 hh = make_float2(h, h);
    u[0] = cadd(e[0], o[0]);
    u[8] = csub(e[0], o[0]);
    u[4] = cadd_mi(e[4], o[4]);
    u[12] = cadd_pi(e[4], o[4]);
    const c32 r2 = cmul_1mi(o[2]), r6 = cmul_1pi(o[6]);
    u[2] = cfma_scale(r2, hh, e[2]);
    u[10] = cfms_scale(r2, hh, e[2]);
    u[6] = cfms_scale(r6, hh, e[6]);
    u[14] = cfma_scale(r6, hh, e[6]);
    const c32 t1 = cmul_pk_sb(o[1], make_float2(c1, -s1)), t3 = cmul_pk_sb(o[3], make_float2(s1, -c1));
    const c32 t5 = cmul_pk_sb(o[5], make_float2(-s1, -c1)), t7 = cmul_pk_sb(o[7], make_float2(-c1, -s1));
    u[1] = cadd(e[1], t1);
    u[9] = csub(e[1], t1);
    u[3] = cadd(e[3], t3);
    u[11] = csub(e[3], t3);
    u[5] = cadd(e[5], t5);
    u[13] = csub(e[5], t5);
    u[7] = cadd(e[7], t7);
    u[15] = csub(e[7], t7);
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
            for (int r = 1; r < R; ++r) v[q * R + r] = cmul_pk(v[q * R + r], tw[q * (R - 1) + r - 1]);
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
// (Two other plans were built, measured and are NOT here: an 8 x 8 register <-> lane transpose in place of one exchange -- 408 against
// 363 ns per transform and SIMD --, and two transforms per wave at 16 points per lane with ONE exchange -- 347 against 363: the exchange halves,
// the butterflies grow by a quarter.  Both live in tools/gpu/kbench/fft_plans.h with their numbers in profiles/r04_k_fft_rate_one_exchange.txt.)
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
// The division by two is NOT done here: the callers transform (a + i b) / 2 -- the factor rides the analysis window
// (load_window_half) or a later weight, exactly (a power of two) -- so A = Z + conj Z', B = -i (Z - conj Z') are one packed
// add each (8 instructions per bin before).
// The partner Z[N - f] of f = lane + 64 j sits in slot E-1-j of lane 64 - lane: it is fetched with one cross-lane read per
// dword (ds_bpermute_b32: the LDS crossbar, no LDS memory, a quarter of the bytes of the write-all / read-pairs round trip
// through the exchange buffer that was here -- the wave transforms are LDS-bandwidth-bound, tools/gpu/kbench/fft_rate.hip).
// Lane 0 is its own partner (f = 64 j pairs with N - 64 j = slot E - j, f = 0 with itself).  Every lane of the wave must call.
template <int N, class Emit>
__device__ __forceinline__ void rfft_pair_untangle(c32* v, c32* buf, int lane, Emit emit) {
    (void)buf;
    constexpr int E = FftPlan<N>::E, EH = E / 2;
    const int src = (64 - lane) & 63;
    c32 zc[EH];
#pragma unroll
    for (int j = 0; j < EH; ++j) {
        const c32 snd = v[E - 1 - j];
        zc[j] = make_float2(__shfl(snd.x, src), __shfl(snd.y, src));
    }
    if (lane == 0) {
        zc[0] = v[0];
#pragma unroll
        for (int j = 1; j < EH; ++j) zc[j] = v[E - j];
    }
#pragma unroll
    for (int j = 0; j < EH; ++j) emit(j, lane + 64 * j, cadd_conj(v[j], zc[j]), csub_conj_mi(v[j], zc[j]));
    if (lane == 0) emit(EH, N / 2, cadd_conj(v[EH], v[EH]), csub_conj_mi(v[EH], v[EH]));
}

// The reverse packing for the inverse transform of two real frames at once: A[j], B[j] = the two spectra at f = lane + 64 j
// (j < E/2; j = E/2: the Nyquist bin, valid in lane 0) -> v[e] = conj(V[lane + 64 e]) with V = A~ + i B~ the sum of the
// Hermitian extensions; FFT(v) then carries frame A in its real and -frame B in its imaginary part (inverse by forward
// transform of the conjugate, scale 1/N left to the caller).  irfft ignores the imaginary parts of DC and Nyquist.
// Upper half n = N - f: conj(V[n]) = conj(conj(A[f]) + i conj(B[f])) = A[f] - i B[f], computed where f lives and fetched from
// lane 64 - lane like the untangle's partner.  Every lane of the wave must call.
template <int N>
__device__ __forceinline__ void irfft_pair_pack(const c32* A, const c32* B, c32* v, int lane) {
    constexpr int E = FftPlan<N>::E, EH = E / 2;
    const int src = (64 - lane) & 63;
    c32 W[EH];
#pragma unroll
    for (int j = 0; j < EH; ++j) {
        v[j] = cadd_pi_conj(A[j], B[j]);         // conj(A + i B)
        W[j] = cadd_mi(A[j], B[j]);              // A - i B
    }
#pragma unroll
    for (int e = EH; e < E; ++e) v[e] = make_float2(__shfl(W[E - 1 - e].x, src), __shfl(W[E - 1 - e].y, src));
    if (lane == 0) {
        v[0] = make_float2(A[0].x - 0.f, -B[0].x);                  // DC: imaginary parts dropped -> conj(A.x + i B.x)
        v[EH] = make_float2(A[EH].x, -B[EH].x);                     // Nyquist likewise
#pragma unroll
        for (int e = EH + 1; e < E; ++e) v[e] = W[E - e];
    }
}

}  // namespace disco
