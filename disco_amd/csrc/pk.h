// Packed-f32 complex arithmetic for gfx950: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with explicit op_sel / neg modifiers.
//
// A complex number lives in an aligned VGPR pair (re = low dword, im = high dword), which is what ds_read_b64 /
// global_load_dwordx2 deliver.  One VOP3P instruction computes both halves of the result; per SOURCE OPERAND its modifiers
// say which half feeds the low result (op_sel), which half feeds the high result (op_sel_hi), and whether that value is
// negated (neg_lo / neg_hi).  With them a conjugation, a multiplication by +-i (swap + one sign) or a broadcast of one half
// costs nothing, and a full complex product is two instructions.  hipcc does form v_pk_* from ext_vector arithmetic and
// folds op_sel, but never the negations (it emits v_xor + v_mov to build the swapped / negated pair first: measured
// slower than scalar code in round 1), so the instructions are written out here, modifiers as template parameters.
//
// Every primitive carries a plain C++ statement of the same modifier semantics, used (a) by the g++ emulator build of the
// kernels (tests/hipemu) -- so the CPU suite checks every modifier choice made in fft.h / k_cov.h against the oracle --
// and (b) by hipcc when DISCO_PKX=0.  disco_selftest_pk (GPU test) runs the instruction forms beside the C++ forms.
#pragma once

#ifndef DISCO_PKX
#define DISCO_PKX 1
#endif

namespace disco {

#if DISCO_PKX && defined(__clang__)
#define DISCO_PKX_ASM 1
#else
#define DISCO_PKX_ASM 0
#endif
#if defined(__clang__)
typedef float pk2f __attribute__((ext_vector_type(2)));
#endif

__device__ __forceinline__ float pk_pick(c32 a, int hi, int neg) {
    const float v = hi ? a.y : a.x;
    return neg ? -v : v;
}

// HW = true: the instruction forms; HW = false: the same modifier semantics in plain C++.
// Template parameters: S?0 / S?1 = which half (0: re, 1: im) of source 0 / 1 feeds the Low / High result;
// N?0 / N?1 = that value negated.
template <bool HW>
struct Pk {
    // r.lo = +-a[SL0] + +-b[SL1],  r.hi = +-a[SH0] + +-b[SH1]
    template <int SL0, int SL1, int SH0, int SH1, int NL0 = 0, int NL1 = 0, int NH0 = 0, int NH1 = 0>
    static __device__ __forceinline__ c32 add(c32 a, c32 b) {
#if defined(__clang__)
        if constexpr (HW) {
            pk2f r;
            const pk2f va = {a.x, a.y}, vb = {b.x, b.y};
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6] neg_lo:[%7,%8] neg_hi:[%9,%10]"
                : "=v"(r)
                : "v"(va), "v"(vb), "n"(SL0), "n"(SL1), "n"(SH0), "n"(SH1), "n"(NL0), "n"(NL1), "n"(NH0), "n"(NH1));
            return make_float2(r.x, r.y);
        }
#endif
        return make_float2(pk_pick(a, SL0, NL0) + pk_pick(b, SL1, NL1), pk_pick(a, SH0, NH0) + pk_pick(b, SH1, NH1));
    }

    // r.lo = (+-a[SL0]) * b[SL1],  r.hi = (+-a[SH0]) * b[SH1].  SB: the multiplier pair b sits in SCALAR registers
    // (compile-time constants: one constant-bus operand per instruction is allowed and no VGPR pair is spent on it)
    template <int SL0, int SL1, int SH0, int SH1, int NL0 = 0, int NH0 = 0, bool SB = false>
    static __device__ __forceinline__ c32 mul(c32 a, c32 b) {
#if defined(__clang__)
        if constexpr (HW) {
            pk2f r;
            const pk2f va = {a.x, a.y}, vb = {b.x, b.y};
            if constexpr (SB)
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6] neg_lo:[%7,0] neg_hi:[%8,0]"
                    : "=v"(r)
                    : "v"(va), "s"(vb), "n"(SL0), "n"(SL1), "n"(SH0), "n"(SH1), "n"(NL0), "n"(NH0));
            else
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[%3,%4] op_sel_hi:[%5,%6] neg_lo:[%7,0] neg_hi:[%8,0]"
                    : "=v"(r)
                    : "v"(va), "v"(vb), "n"(SL0), "n"(SL1), "n"(SH0), "n"(SH1), "n"(NL0), "n"(NH0));
            return make_float2(r.x, r.y);
        }
#endif
        return make_float2(pk_pick(a, SL0, NL0) * pk_pick(b, SL1, 0), pk_pick(a, SH0, NH0) * pk_pick(b, SH1, 0));
    }

    // r.lo = fma(+-a[SL0], b[SL1], c.lo),  r.hi = fma(+-a[SH0], b[SH1], c.hi)
    template <int SL0, int SL1, int SH0, int SH1, int NL0 = 0, int NH0 = 0, bool SB = false>
    static __device__ __forceinline__ c32 fma(c32 a, c32 b, c32 c) {
#if defined(__clang__)
        if constexpr (HW) {
            pk2f r;
            const pk2f va = {a.x, a.y}, vb = {b.x, b.y}, vc = {c.x, c.y};
            if constexpr (SB)
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[%4,%5,0] op_sel_hi:[%6,%7,1] neg_lo:[%8,0,0] neg_hi:[%9,0,0]"
                    : "=v"(r)
                    : "v"(va), "s"(vb), "v"(vc), "n"(SL0), "n"(SL1), "n"(SH0), "n"(SH1), "n"(NL0), "n"(NH0));
            else
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[%4,%5,0] op_sel_hi:[%6,%7,1] neg_lo:[%8,0,0] neg_hi:[%9,0,0]"
                    : "=v"(r)
                    : "v"(va), "v"(vb), "v"(vc), "n"(SL0), "n"(SL1), "n"(SH0), "n"(SH1), "n"(NL0), "n"(NH0));
            return make_float2(r.x, r.y);
        }
#endif
        return make_float2(fmaf(pk_pick(a, SL0, NL0), pk_pick(b, SL1, 0), c.x), fmaf(pk_pick(a, SH0, NH0), pk_pick(b, SH1, 0), c.y));
    }

    // ---- the complex operations built from them -----------------------------------------------------------------
    // a + conj(b)
    static __device__ __forceinline__ c32 cadd_conj(c32 a, c32 b) { return add<0, 0, 1, 1, 0, 0, 0, 1>(a, b); }
    // -i (a - conj(b)) = (a.y + b.y, b.x - a.x)
    static __device__ __forceinline__ c32 csub_conj_mi(c32 a, c32 b) { return add<1, 1, 0, 0, 0, 0, 1, 0>(a, b); }
    // a + (-i) b = (a.x + b.y, a.y - b.x)
    static __device__ __forceinline__ c32 cadd_mi(c32 a, c32 b) { return add<0, 1, 1, 0, 0, 0, 0, 1>(a, b); }
    // a + i b = (a.x - b.y, a.y + b.x)
    static __device__ __forceinline__ c32 cadd_pi(c32 a, c32 b) { return add<0, 1, 1, 0, 0, 1, 0, 0>(a, b); }
    // conj(a + i b) = (a.x - b.y, -a.y - b.x)
    static __device__ __forceinline__ c32 cadd_pi_conj(c32 a, c32 b) { return add<0, 1, 1, 0, 0, 1, 1, 1>(a, b); }
    // (1 - i) a = (a.x + a.y, a.y - a.x)
    static __device__ __forceinline__ c32 cmul_1mi(c32 a) { return add<0, 1, 1, 0, 0, 0, 0, 1>(a, a); }
    // (1 + i) a = (a.x - a.y, a.x + a.y)
    static __device__ __forceinline__ c32 cmul_1pi(c32 a) { return add<0, 1, 0, 1, 0, 1, 0, 0>(a, a); }
    // a t = (a.x t.x - a.y t.y, a.x t.y + a.y t.x); SB: t a compile-time constant
    template <bool SB = false>
    static __device__ __forceinline__ c32 cmul(c32 a, c32 t) {
        const c32 r = mul<0, 0, 0, 1, 0, 0, SB>(a, t);                 // (a.x t.x, a.x t.y)
        return fma<1, 1, 1, 0, 1, 0, SB>(a, t, r);                     // (-a.y t.y + ., a.y t.x + .)
    }
    // acc + a t = (acc.x + a.x t.x - a.y t.y, acc.y + a.x t.y + a.y t.x)
    static __device__ __forceinline__ c32 cfma(c32 a, c32 t, c32 acc) {
        const c32 r = fma<0, 0, 0, 1>(a, t, acc);                      // (a.x t.x + ., a.x t.y + .)
        return fma<1, 1, 1, 0, 1, 0>(a, t, r);                         // (-a.y t.y + ., a.y t.x + .)
    }
    // a conj(b) = (a.x b.x + a.y b.y, a.y b.x - a.x b.y)
    static __device__ __forceinline__ c32 cmul_aconjb(c32 a, c32 b) {
        const c32 r = mul<0, 0, 1, 0>(a, b);                           // (a.x b.x, a.y b.x)
        return fma<1, 1, 0, 1, 0, 1>(a, b, r);                         // (a.y b.y + ., -a.x b.y + .)
    }
    // acc + conj(w) x = (acc.x + w.x x.x + w.y x.y, acc.y + w.x x.y - w.y x.x)
    static __device__ __forceinline__ c32 cfma_conj(c32 w, c32 x, c32 acc) {
        const c32 r = fma<0, 0, 0, 1>(w, x, acc);                      // (w.x x.x + ., w.x x.y + .)
        return fma<1, 1, 1, 0, 0, 1>(w, x, r);                         // (w.y x.y + ., -w.y x.x + .)
    }
    // e + s h and e - s h (component-wise), hh = (h, h) a compile-time constant pair
    static __device__ __forceinline__ c32 cfma_scale(c32 s, c32 hh, c32 e) { return fma<0, 0, 1, 1, 0, 0, true>(s, hh, e); }
    static __device__ __forceinline__ c32 cfms_scale(c32 s, c32 hh, c32 e) { return fma<0, 0, 1, 1, 1, 1, true>(s, hh, e); }
    // component-wise: (a.x b.x, a.y b.y) and (c.x + a.x b.x, c.y + a.y b.y)  (a real diagonal held two to a pair; |z|^2 as two squares)
    static __device__ __forceinline__ c32 mul_comp(c32 a, c32 b) { return mul<0, 0, 1, 1>(a, b); }
    static __device__ __forceinline__ c32 fma_comp(c32 a, c32 b, c32 c) { return fma<0, 0, 1, 1>(a, b, c); }
    // the two halves of acc + u v and of acc + conj(u) v as separate instructions, for callers that advance many sums together
    // (k_solve_small.h): lo = (u.x v.x + ., u.x v.y + .), hi = (-u.y v.y + ., +u.y v.x + .), hi_conj = (+u.y v.y + ., -u.y v.x + .)
    static __device__ __forceinline__ c32 cfma_lo(c32 u, c32 v, c32 acc) { return fma<0, 0, 0, 1>(u, v, acc); }
    static __device__ __forceinline__ c32 cfma_hi(c32 u, c32 v, c32 acc) { return fma<1, 1, 1, 0, 1, 0>(u, v, acc); }
    static __device__ __forceinline__ c32 cfma_hi_conj(c32 u, c32 v, c32 acc) { return fma<1, 1, 1, 0, 0, 1>(u, v, acc); }
    // (a.x w[S], a.y w[S]): both halves weighted by one half of the pair w
    template <int S>
    static __device__ __forceinline__ c32 scale_by_half(c32 a, c32 w) { return mul<0, S, 1, S>(a, w); }
    // acc + p w[S] (component-wise)
    template <int S>
    static __device__ __forceinline__ c32 fma_by_half(c32 p, c32 w, c32 acc) { return fma<0, S, 1, S>(p, w, acc); }
};

typedef Pk<DISCO_PKX_ASM != 0> PkD;
__device__ __forceinline__ c32 cadd_conj(c32 a, c32 b) { return PkD::cadd_conj(a, b); }
__device__ __forceinline__ c32 csub_conj_mi(c32 a, c32 b) { return PkD::csub_conj_mi(a, b); }
__device__ __forceinline__ c32 cadd_mi(c32 a, c32 b) { return PkD::cadd_mi(a, b); }
__device__ __forceinline__ c32 cadd_pi(c32 a, c32 b) { return PkD::cadd_pi(a, b); }
__device__ __forceinline__ c32 cadd_pi_conj(c32 a, c32 b) { return PkD::cadd_pi_conj(a, b); }
__device__ __forceinline__ c32 cmul_1mi(c32 a) { return PkD::cmul_1mi(a); }
__device__ __forceinline__ c32 cmul_1pi(c32 a) { return PkD::cmul_1pi(a); }
__device__ __forceinline__ c32 cmul_pk(c32 a, c32 t) { return PkD::cmul<false>(a, t); }
__device__ __forceinline__ c32 cmul_pk_sb(c32 a, c32 t) { return PkD::cmul<true>(a, t); }
__device__ __forceinline__ c32 cmul_aconjb(c32 a, c32 b) { return PkD::cmul_aconjb(a, b); }
__device__ __forceinline__ c32 cfma_conj(c32 w, c32 x, c32 acc) { return PkD::cfma_conj(w, x, acc); }
__device__ __forceinline__ c32 cfma_pk(c32 a, c32 t, c32 acc) { return PkD::cfma(a, t, acc); }
__device__ __forceinline__ c32 cfma_scale(c32 s, c32 hh, c32 e) { return PkD::cfma_scale(s, hh, e); }
__device__ __forceinline__ c32 cfms_scale(c32 s, c32 hh, c32 e) { return PkD::cfms_scale(s, hh, e); }
template <int S>
__device__ __forceinline__ c32 scale_by_half(c32 a, c32 w) { return PkD::template scale_by_half<S>(a, w); }
template <int S>
__device__ __forceinline__ c32 fma_by_half(c32 p, c32 w, c32 acc) { return PkD::template fma_by_half<S>(p, w, acc); }

// Self-test (disco_selftest_pk): every operation above through the instruction forms (out_hw) and through the C++ forms
// (out_ref) on the same operands; the GPU test demands bit equality of the two and agreement with complex arithmetic in NumPy.
constexpr int PK_SELFTEST_OPS = 23;
template <bool HW>
__device__ __forceinline__ void pk_selftest_ops(c32 a, c32 b, c32 c, c32* o) {
    const c32 hh = make_float2(0.70710678118654752440f, 0.70710678118654752440f);
    const c32 kt = make_float2(0.92387953251128675613f, -0.38268343236508977173f);
    o[0] = Pk<HW>::cadd_conj(a, b);
    o[1] = Pk<HW>::csub_conj_mi(a, b);
    o[2] = Pk<HW>::cadd_mi(a, b);
    o[3] = Pk<HW>::cadd_pi(a, b);
    o[4] = Pk<HW>::cadd_pi_conj(a, b);
    o[5] = Pk<HW>::cmul_1mi(a);
    o[6] = Pk<HW>::cmul_1pi(a);
    o[7] = Pk<HW>::template cmul<false>(a, b);
    o[8] = Pk<HW>::template cmul<true>(a, kt);
    o[9] = Pk<HW>::cmul_aconjb(a, b);
    o[10] = Pk<HW>::cfma_conj(a, b, c);
    o[11] = Pk<HW>::cfma_scale(a, hh, c);
    o[12] = Pk<HW>::cfms_scale(a, hh, c);
    o[13] = Pk<HW>::template scale_by_half<0>(a, b);
    o[14] = Pk<HW>::template scale_by_half<1>(a, b);
    o[15] = Pk<HW>::template fma_by_half<1>(a, b, c);
    o[16] = Pk<HW>::cfma(a, b, c);
    o[17] = Pk<HW>::mul_comp(a, b);
    o[18] = Pk<HW>::fma_comp(a, b, c);
    o[19] = Pk<HW>::cfma_lo(a, b, c);
    o[20] = Pk<HW>::cfma_hi(a, b, c);
    o[21] = Pk<HW>::cfma_hi_conj(a, b, c);
    o[22] = Pk<HW>::template fma_by_half<0>(a, b, c);
}
static __global__ void k_pk_selftest(const c32* __restrict__ a, const c32* __restrict__ b, const c32* __restrict__ c, long long n,
                              c32* __restrict__ out_hw, c32* __restrict__ out_ref) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    c32 o[PK_SELFTEST_OPS];
    pk_selftest_ops<true>(a[i], b[i], c[i], o);
#pragma unroll
    for (int q = 0; q < PK_SELFTEST_OPS; ++q) out_hw[i * PK_SELFTEST_OPS + q] = o[q];
    pk_selftest_ops<false>(a[i], b[i], c[i], o);
#pragma unroll
    for (int q = 0; q < PK_SELFTEST_OPS; ++q) out_ref[i * PK_SELFTEST_OPS + q] = o[q];
}

}  // namespace disco
