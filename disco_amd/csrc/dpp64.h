// float64 arithmetic against a value held by ANOTHER lane of the same 16-lane row, without LDS: gfx950's DPP control
// `row_newbcast:n` (lane n of every row of 16 as src0) on the two 64-bit VOP2 forms that take it, v_fmac_f64 and v_mov_b64.
// Measured on the MI355X (tools/gpu/kbench/dpp64_rate.hip): v_fmac_f64_dpp issues at the rate of the plain v_fma_f64
// (2.0 ns per wave64 instruction and SIMD, dependent chains included); the LDS form it replaces -- one broadcast ds_read_b128
// per complex multiply-add -- runs at 3.8 ns per v_fma_f64 with every SIMD of a CU reading.
//
// A group of 16 lanes that owns one small dense problem (lane j = column j, k_solve_dpp.h) reads "entry i of lane k's column"
// as `row_newbcast:k` on the register that holds entry i: the register index is the same in every lane, the lane index is an
// immediate.  What DPP cannot do is read a register whose INDEX depends on the receiving lane (a transposition): those go
// through LDS once.
//
// Hazards the compiler cannot see inside inline asm (GCNHazardRecognizer::checkDPPHazards): a VGPR written by a VALU
// instruction must not be read through DPP within the next 2 wait states, and EXEC must not have been written within the last 5.
// Every helper below that starts a run of DPP reads first pins the values it is about to read (an empty asm that takes them as
// inputs: they are computed BEFORE this point) and then issues s_nop 1; code that leaves a divergent region calls
// DISCO_DPP_SETTLE() before the next helper.  disco_amd/build.py checks both rules on the generated ISA of every kernel that
// uses these forms (check_dpp_hazards).
//
// The g++ emulator build (tests/hipemu) has no lanes to read from: the same helpers fetch the other lane's values through the
// emulator's wave exchange area and run the same fused multiply-adds in the same order, so results are bit-identical.
#pragma once
#include "common.h"

#include <type_traits>

namespace disco {

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// sign pattern of a complex multiply-add against a broadcast value S (src0, through DPP) and an own value o:
//   acc.x += (n0 ? -1 : 1) S.x o.x + (n1 ? -1 : 1) S.y o.y;   acc.y += (n2 ? -1 : 1) S.x o.y + (n3 ? -1 : 1) S.y o.x
enum ZMode {
    Z_ADD_SO = 0x4,      // acc += S o                 (+, -, +, +)
    Z_SUB_SO = 0xB,      // acc -= S o                 (-, +, -, -)
    Z_SUB_OCS = 0xE,     // acc -= o conj(S)           (-, -, -, +)
    Z_ADD_COS = 0x2,     // acc += conj(o) S           (+, +, -, +)
    Z_SUB_COS = 0xD,     // acc -= conj(o) S           (-, -, +, -)
};

#if defined(__clang__)
#define DISCO_DPP_SETTLE() asm volatile("s_nop 4")
__device__ __forceinline__ void dpp_pin(const double& a, const double& b) { asm volatile("" ::"v"(a), "v"(b)); }
__device__ __forceinline__ void dpp_pin(const c64& a) { asm volatile("" ::"v"(a.x), "v"(a.y)); }
#define DISCO_DPP_SOURCES_READY() asm volatile("s_nop 1")

template <int K>
__device__ __forceinline__ double mov_bc(const double& s) {
    double r;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(s), "n"(K));
    return r;
}
// the four multiply-adds of one complex multiply-add as ONE asm statement: between two asm statements of which the second reads a
// register the first wrote, hipcc inserts `s_nop 0` (it cannot see inside them; gfx950's forwarding hazards are assumed) -- 437 of
// the 1 601 instructions of a P = 15 squaring before this.  The hardware interlocks the accumulator chain itself.
#define DISCO_ZFMA_ASM(N0, N1, N2, N3)                                                                      \
    asm volatile("v_fmac_f64_dpp %0, " N0 "%2, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n"             \
                 "v_fmac_f64_dpp %0, " N1 "%3, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n"             \
                 "v_fmac_f64_dpp %1, " N2 "%2, %5 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n"             \
                 "v_fmac_f64_dpp %1, " N3 "%3, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf"                \
                 : "+v"(acc.x), "+v"(acc.y)                                                                 \
                 : "v"(S.x), "v"(S.y), "v"(o.x), "v"(o.y), "n"(K))
template <int K, int MODE>
__device__ __forceinline__ void zfma_bc(c64& acc, const c64& S, const c64& o) {
    static_assert(MODE == Z_ADD_SO || MODE == Z_SUB_SO || MODE == Z_SUB_OCS || MODE == Z_ADD_COS || MODE == Z_SUB_COS, "sign pattern");
    if constexpr (MODE == Z_ADD_SO) DISCO_ZFMA_ASM("", "-", "", "");
    if constexpr (MODE == Z_SUB_SO) DISCO_ZFMA_ASM("-", "", "-", "-");
    if constexpr (MODE == Z_SUB_OCS) DISCO_ZFMA_ASM("-", "-", "-", "");
    if constexpr (MODE == Z_ADD_COS) DISCO_ZFMA_ASM("", "", "-", "");
    if constexpr (MODE == Z_SUB_COS) DISCO_ZFMA_ASM("-", "-", "", "-");
}
// sum over the 16 lanes of the row, in lane order (lanes that do not take part must hold 0): one statement, see above
__device__ __forceinline__ double sum16_bc(const double& v) {
    double r;
    const double one = 1.0;
#define DISCO_S16(K) "v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf\n"
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" DISCO_S16(1) DISCO_S16(2) DISCO_S16(3) DISCO_S16(4) DISCO_S16(5)
                     DISCO_S16(6) DISCO_S16(7) DISCO_S16(8) DISCO_S16(9) DISCO_S16(10) DISCO_S16(11) DISCO_S16(12) DISCO_S16(13) DISCO_S16(14)
                         "v_fmac_f64_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "=&v"(r)
                 : "v"(v), "v"(one));
#undef DISCO_S16
    return r;
}
#else
#define DISCO_DPP_SETTLE() ((void)0)
#endif

// the same multiply-add with S already in hand (emulator build; also the statement of what zfma_bc computes)
template <int MODE>
__device__ __forceinline__ void zfma_plain(c64& acc, const c64& S, const c64& o) {
    acc.x = fma((MODE & 8) ? -S.x : S.x, o.x, acc.x);
    acc.x = fma((MODE & 4) ? -S.y : S.y, o.y, acc.x);
    acc.y = fma((MODE & 2) ? -S.x : S.x, o.y, acc.y);
    acc.y = fma((MODE & 1) ? -S.y : S.y, o.x, acc.y);
}

// The first N entries of lane K's register array `a` (same array, same indices, in every lane).  The array must not change while
// the view is in use.
// SETTLED: the caller vouches that the entries were pinned and settled by an earlier view and not written since (the k-loop of a squaring).
template <int K, int N, int NA, bool SETTLED = false>
struct BcRow {
#if defined(__clang__)
    const c64 (&src)[NA];
    __device__ __forceinline__ explicit BcRow(const c64 (&a)[NA]) : src(a) {
        if constexpr (!SETTLED) {
#pragma unroll
            for (int i = 0; i < N; ++i) dpp_pin(a[i]);
            DISCO_DPP_SOURCES_READY();
        }
    }
    template <int MODE>
    __device__ __forceinline__ void fma(c64& acc, int i, const c64& o) const { zfma_bc<K, MODE>(acc, src[i], o); }
    __device__ __forceinline__ double re(int i) const { return mov_bc<K>(src[i].x); }
#else
    c64 tmp[N];
    explicit BcRow(const c64 (&a)[NA]) { hipemu::gather_lane(a, sizeof(c64) * N, (hipemu::t_lane & ~15) + K, tmp); }
    template <int MODE>
    void fma(c64& acc, int i, const c64& o) const { zfma_plain<MODE>(acc, tmp[i], o); }
    double re(int i) const { return tmp[i].x; }
#endif
};

// One complex value per lane, read from any lane of the row.
struct BcVec {
#if defined(__clang__)
    const c64 v;
    __device__ __forceinline__ explicit BcVec(const c64& x) : v(x) {
        dpp_pin(v);
        DISCO_DPP_SOURCES_READY();
    }
    template <int K, int MODE>
    __device__ __forceinline__ void fma(c64& acc, const c64& o) const { zfma_bc<K, MODE>(acc, v, o); }
    template <int K>
    __device__ __forceinline__ c64 get() const { return make_double2(mov_bc<K>(v.x), mov_bc<K>(v.y)); }
#else
    c64 all[16];
    explicit BcVec(const c64& x) { hipemu::gather_row16(&x, sizeof(c64), all); }
    template <int K, int MODE>
    void fma(c64& acc, const c64& o) const { zfma_plain<MODE>(acc, all[K], o); }
    template <int K>
    c64 get() const { return all[K]; }
#endif
};

// One double per lane, read from any lane of the row; sum over the 16 lanes of the row in lane order (every lane gets the same
// bits; lanes that do not take part must hold 0).
struct BcReal {
#if defined(__clang__)
    const double v;
    __device__ __forceinline__ explicit BcReal(const double& x) : v(x) {
        dpp_pin(v, v);
        DISCO_DPP_SOURCES_READY();
    }
    template <int K>
    __device__ __forceinline__ double get() const { return mov_bc<K>(v); }
    __device__ __forceinline__ double sum() const { return sum16_bc(v); }
#else
    double all[16];
    explicit BcReal(const double& x) { hipemu::gather_row16(&x, sizeof(double), all); }
    template <int K>
    double get() const { return all[K]; }
    double sum() const {
        double r = all[0];
        for (int k = 1; k < 16; ++k) r = fma(all[k], 1.0, r);
        return r;
    }
#endif
};

// ---- self-test (disco_selftest_dpp): every helper above through its instruction form and through __shfl + the plain statement
constexpr int DPP_SELFTEST_OPS = 8;
template <bool HW>
__device__ __forceinline__ void dpp_selftest_ops(const c64 a, const c64 b, c64* o) {
    auto via_shfl = [&](int k, c64 v) { return make_double2(__shfl(v.x, k, 16), __shfl(v.y, k, 16)); };
#pragma unroll
    for (int q = 0; q < 5; ++q) o[q] = a;
    if constexpr (HW) {
        const BcVec bv(b);
        bv.template fma<0, Z_ADD_SO>(o[0], a);
        bv.template fma<5, Z_SUB_SO>(o[1], a);
        bv.template fma<9, Z_SUB_OCS>(o[2], a);
        bv.template fma<15, Z_ADD_COS>(o[3], a);
        bv.template fma<3, Z_SUB_COS>(o[4], a);
        o[5] = make_double2(BcReal(b.x).template get<7>(), BcReal(b.y).template get<12>());
        o[6] = make_double2(BcReal(a.x).sum(), BcReal(b.y).sum());
        const c64 arr[2] = {a, b};
        const BcRow<6, 2, 2> row(arr);
        o[7] = make_double2(row.re(0), 0.0);
        row.template fma<Z_ADD_SO>(o[7], 1, a);
    } else {
        zfma_plain<Z_ADD_SO>(o[0], via_shfl(0, b), a);
        zfma_plain<Z_SUB_SO>(o[1], via_shfl(5, b), a);
        zfma_plain<Z_SUB_OCS>(o[2], via_shfl(9, b), a);
        zfma_plain<Z_ADD_COS>(o[3], via_shfl(15, b), a);
        zfma_plain<Z_SUB_COS>(o[4], via_shfl(3, b), a);
        o[5] = make_double2(__shfl(b.x, 7, 16), __shfl(b.y, 12, 16));
        double sa = __shfl(a.x, 0, 16), sb = __shfl(b.y, 0, 16);
        for (int k = 1; k < 16; ++k) {
            sa = fma(__shfl(a.x, k, 16), 1.0, sa);
            sb = fma(__shfl(b.y, k, 16), 1.0, sb);
        }
        o[6] = make_double2(sa, sb);
        o[7] = make_double2(__shfl(a.x, 6, 16), 0.0);
        zfma_plain<Z_ADD_SO>(o[7], via_shfl(6, b), a);
    }
}
static __global__ __launch_bounds__(64) void k_dpp_selftest(const c64* __restrict__ a, const c64* __restrict__ b, c64* __restrict__ out_hw,
                                                            c64* __restrict__ out_ref) {
    const long long i = (long long)blockIdx.x * 64 + threadIdx.x;
    c64 o[DPP_SELFTEST_OPS];
    dpp_selftest_ops<true>(a[i], b[i], o);
#pragma unroll
    for (int q = 0; q < DPP_SELFTEST_OPS; ++q) out_hw[i * DPP_SELFTEST_OPS + q] = o[q];
    dpp_selftest_ops<false>(a[i], b[i], o);
#pragma unroll
    for (int q = 0; q < DPP_SELFTEST_OPS; ++q) out_ref[i * DPP_SELFTEST_OPS + q] = o[q];
}

}  // namespace disco
