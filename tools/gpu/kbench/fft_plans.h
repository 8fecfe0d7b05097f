// Wave-transform plans that were built, checked and MEASURED SLOWER OR NO BETTER than csrc/fft.h's fft_wave (three radix-8 passes, two
// LDS exchanges) -- kept out of the product, here as the record (tools/gpu/kbench/fft_rate.hip times them):
//   fft_wave_xlane   (round 2) one exchange replaced by an 8 x 8 register <-> lane transpose: 408 against 363 ns per transform and SIMD
//   fft_wave_2x512   (round 4) two transforms per wave, 16 points per lane, ONE exchange: 347 against 363 ns (4 waves / SIMD), 364 against
//                    391 (3 waves): the exchange costs 122 instead of 276 ns, the butterflies + twiddles + lane transposes 312 instead of 243
//                    (profiles/r04_k_fft_rate_one_exchange.txt) -- 5 % for a re-plumbing of every transform kernel: not taken.
#pragma once
#include "../../../disco_amd/csrc/fft.h"

namespace disco {

// ---- 512 points with ONE trip through LDS ------------------------------------------------------------------------------
// Measured on the MI355X (tools/gpu/kbench/fft_rate.hip): the two LDS exchanges of the Stockham schedule below cost 275 ns per
// transform and SIMD, the butterflies 250 ns -- the wave transforms are LDS-BANDWIDTH-bound (118 of the CU's 128 B/clk).
// fft_wave_xlane runs the same three radix-8 passes as a decimation in frequency on n = 64 n2 + 8 n1 + n0 (slot = n2,
// lane = 8 n1 + n0), k = k0 + 8 k1 + 64 k2:
//   pass 1 over the slots (n2 -> k0), twiddle W_512^(k0 lane);
//   8 x 8 transpose between the slot index and lane bits 5:3 WITHOUT LDS: v_permlane32_swap / v_permlane16_swap (the 2 x 2 block
//   transposes of gfx950) for lane bits 5 and 4, a DPP row rotate by 8 with bank masks for lane bit 3  (slot = n1, lane = 8 k0 + n0);
//   pass 2 over the slots (n1 -> k1), twiddle W_64^(k1 n0);
//   one LDS exchange that also undoes the digit order (lane = k0 + 8 k1, slot = n0);
//   pass 3 over the slots (n0 -> k2): lane holds X[lane + 64 slot], the natural order of fft_wave.
// MEASURED SLOWER and therefore off: 408 ns per transform and SIMD at 4 waves/SIMD against 363 ns for the Stockham schedule
// (442 vs 390 at 3 waves): the 16 permlane swaps + 16 DPP moves cost more VALU time than the LDS exchange they replace
// frees.  Kept (checked against the oracle on the emulated build, and on the MI355X by fft_rate's check) as the record of it.

// 2 x 2 transpose between two registers and lane bit BIT (8, 16 or 32): lanes with the bit set receive the partner's b in a,
// lanes with it clear receive the partner's a in b (partner = lane ^ BIT).
template <int BIT>
__device__ __forceinline__ void xlane_swap(float& a, float& b, int lane) {
#if defined(__clang__)
    (void)lane;
    if constexpr (BIT == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]);
        b = __uint_as_float(r[1]);
    } else if constexpr (BIT == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
        a = __uint_as_float(r[0]);
        b = __uint_as_float(r[1]);
    } else {
        static_assert(BIT == 8, "lane bit");
        const int ai = __float_as_int(a), bi = __float_as_int(b);
        // row_ror:8 (0x128): lane i of a 16-lane row reads lane i - 8 (mod 16); bank_mask picks lanes 8..15 (0xc) / 0..7 (0x3)
        a = __int_as_float(__builtin_amdgcn_update_dpp(ai, bi, 0x128, 0xf, 0xc, false));
        b = __int_as_float(__builtin_amdgcn_update_dpp(bi, ai, 0x128, 0xf, 0x3, false));
    }
#else
    const float pa = __shfl_xor(a, BIT), pb = __shfl_xor(b, BIT);
    if (lane & BIT) a = pb;
    else b = pa;
#endif
}
template <int BIT>
__device__ __forceinline__ void xlane_swap(c32& a, c32& b, int lane) {
    xlane_swap<BIT>(a.x, b.x, lane);
    xlane_swap<BIT>(a.y, b.y, lane);
}
// slot index <-> lane bits 5:3
__device__ __forceinline__ void xlane_transpose8(c32* v, int lane) {
#pragma unroll
    for (int s = 0; s < 4; ++s) xlane_swap<32>(v[s], v[s + 4], lane);
#pragma unroll
    for (int s = 0; s < 8; ++s)
        if ((s & 2) == 0) xlane_swap<16>(v[s], v[s + 2], lane);
#pragma unroll
    for (int s = 0; s < 8; s += 2) xlane_swap<8>(v[s], v[s + 1], lane);
}

struct WaveTwXlane {
    c32 t1[7], t2[7];
    __device__ __forceinline__ void init(const c32* __restrict__ tw, int lane) {
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            t1[r - 1] = tw[r * lane];                   // W_512^(k0 (8 n1 + n0))
            t2[r - 1] = tw[8 * r * (lane & 7)];         // W_64^(k1 n0)
        }
    }
};
// Forward complex FFT of the wave's N points.  In: v[e] = x[lane + 64 e].  Out: v[e] = X[lane + 64 e].
// `buf` = wave-private LDS of fft_buf_len<N>() c32.
__device__ __forceinline__ void fft_wave_xlane(c32* v, const WaveTwXlane& tw, c32* buf, int lane) {
    constexpr int N = 512;
    dft8(v);                                                     // n2 -> k0
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cmul_pk(v[r], tw.t1[r - 1]);
    xlane_transpose8(v, lane);                                   // slot = n1, lane = 8 k0 + n0
    dft8(v);                                                     // n1 -> k1
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cmul_pk(v[r], tw.t2[r - 1]);
    DISCO_LDS_WAR();        // the previous user of `buf` is done in every lane
    const int w0 = (lane >> 3) + 64 * (lane & 7);                // element (k0, n0; k1 = r) -> position k0 + 8 k1 + 64 n0
#pragma unroll
    for (int r = 0; r < 8; ++r) buf[fft_pad<N>(w0 + 8 * r)] = v[r];
    DISCO_LDS_RAW();
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = buf[fft_pad<N>(lane + 64 * e)];
    DISCO_LDS_WAR();
    dft8(v);                                                     // n0 -> k2: v[e] = X[lane + 64 e]
}

// ---- TWO 512-point transforms per wave with ONE trip through LDS (round 4 experiment) ---------------------
// Half-wave h (lanes 32 h ... 32 h + 31) owns transform h with 16 points per lane: 512 = 16 x 2 x 16, decimation in frequency on
// n = n0 + 32 n1 (lane l' = n0, slot = n1), k = k1 + 16 (2 m + c):
//   radix 16 over the slots (n1 -> k1), twiddle W_512^(n0 k1) -- the 15 factors are products of four per-lane registers W^(n0), W^(2 n0),
//   W^(4 n0), W^(8 n0) --, radix 2 across lane bit 4 (n0 = j + 16 hh) by eight 2 x 2 register / lane transposes (v_permlane16_swap: the
//   halves swap slots s and s + 8, every lane then forms sums AND differences of eight slots), the differences times W_32^j (one register),
//   ONE LDS exchange (lane j writes its (k1, c) values into row k1 + 16 c, lane l' reads row l'), radix 16 over j (-> m).
// In: v[s] = x_h[l' + 32 s].  Out: v[e] = X_h[l' + 32 e] -- the natural order of fft_wave, per half-wave.
// Against fft_wave<512> per transform: the same butterflies + 10 % (117 packed instructions instead of 106), HALF the LDS traffic
// (one exchange of 16 values per lane for two transforms).  `buf`: wave-private, FFT2X_BUF c32.
constexpr int FFT2X_PITCH = 18;                         // row pitch (c32): 16-byte aligned rows, the rows of the two halves of a write on different banks
constexpr int FFT2X_BUF = 2 * 32 * FFT2X_PITCH;
struct WaveTw2x {
    c32 w1, w2, w4, w8, r;                              // W_512^(n0), ^(2 n0), ^(4 n0), ^(8 n0), n0 = lane & 31;  W_32^(lane & 15)
    __device__ __forceinline__ void init(const c32* __restrict__ tw512, int lane) {
        const int n0 = lane & 31;
        w1 = tw512[n0];
        w2 = tw512[2 * n0];
        w4 = tw512[4 * n0];
        w8 = tw512[8 * n0];
        r = tw512[16 * (lane & 15)];
    }
};
__device__ __forceinline__ void fft_wave_2x512(c32* v, const WaveTw2x& tw, c32* buf, int lane) {
    dft16(v);                                                    // n1 -> k1
    {   // v[k] *= W_512^(n0 k): binary products of the four kept powers
        const c32 t3 = cmul_pk(tw.w1, tw.w2);
        v[1] = cmul_pk(v[1], tw.w1);
        v[2] = cmul_pk(v[2], tw.w2);
        v[3] = cmul_pk(v[3], t3);
        v[4] = cmul_pk(v[4], tw.w4);
        const c32 t5 = cmul_pk(tw.w4, tw.w1), t6 = cmul_pk(tw.w4, tw.w2), t7 = cmul_pk(tw.w4, t3);
        v[5] = cmul_pk(v[5], t5);
        v[6] = cmul_pk(v[6], t6);
        v[7] = cmul_pk(v[7], t7);
        v[8] = cmul_pk(v[8], tw.w8);
        v[9] = cmul_pk(v[9], cmul_pk(tw.w8, tw.w1));
        v[10] = cmul_pk(v[10], cmul_pk(tw.w8, tw.w2));
        v[11] = cmul_pk(v[11], cmul_pk(tw.w8, t3));
        v[12] = cmul_pk(v[12], cmul_pk(tw.w8, tw.w4));
        v[13] = cmul_pk(v[13], cmul_pk(tw.w8, t5));
        v[14] = cmul_pk(v[14], cmul_pk(tw.w8, t6));
        v[15] = cmul_pk(v[15], cmul_pk(tw.w8, t7));
    }
    // radix 2 across lane bit 4: after the transposes a lane holds (element n0 = j, element n0 = j + 16) of slots k1 = s + 8 hh
    const int hh = (lane >> 4) & 1, j = lane & 15, half = lane >> 5;
    DISCO_LDS_WAR();                                             // the previous user of `buf` is done in every lane
    c32* rows = buf + half * (32 * FFT2X_PITCH);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        c32 a = v[s], b = v[s + 8];
        xlane_swap<16>(a, b, lane);
        const c32 sum = cadd(a, b), dif = cmul_pk(csub(a, b), tw.r);
        rows[(s + 8 * hh) * FFT2X_PITCH + j] = sum;              // c = 0: row k1
        rows[(s + 8 * hh + 16) * FFT2X_PITCH + j] = dif;         // c = 1: row k1 + 16
    }
    DISCO_LDS_RAW();
    const float4* rd = reinterpret_cast<const float4*>(rows + (lane & 31) * FFT2X_PITCH);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 t = rd[q];
        v[2 * q] = make_float2(t.x, t.y);
        v[2 * q + 1] = make_float2(t.z, t.w);
    }
    DISCO_LDS_WAR();
    dft16(v);                                                    // j -> m: v[m] = X[l' + 32 m]
}

}  // namespace disco
