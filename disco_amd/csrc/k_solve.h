// Batched rank-1 GEVD-MWF solve -- intern_filter(Rxx, Rnn, mu, type='gevd', rank=1),
// disco_theque/se_utils/internal_formulas.py:56-73, for Hermitian pencils of size P <= 16.
//
// The reference calls LAPACK's non-symmetric complex64 generalized eigensolver and forms
//     w = Q D (D + mu I)^-1 Q^-1 [:, 0]  (D keeps only the largest, clamped eigenvalue),  t1 = Q[:,0] (Q^-1)[0,0].
// Both are invariant to the scale/phase of the top eigenvector q0, and for a Hermitian pencil with
// Rnn = L L^H,  C = L^-1 Rxx L^-H = V diag(d) V^H  one has Q = L^-H V, Q^-1 = V^H L^H, hence
//     t1 = L^-H v0 * L[0,0] * conj(v0[0]),     w = t1 * d0 / (d0 + mu),   d0 clamped to [eps, 1e6]
// (oracle/mwf_oracle.py:gevd_mwf_r1_hermitian; checked against the reference's own outputs).
//
// (9 <= P <= 16 of the batch solve run on the register / DPP form of the same algorithm by default: k_solve_dpp.h; this file is the
// LDS form -- P = 5 ... 8, the online kernel, k_mwf_variants, and option "solve_dpp" = 0.)
// Mapping: a group of G = 4 / 8 / 16 lanes owns one problem, lane j owns column j.  Everything is float64:
// cooperative Cholesky through LDS, two forward substitutions (column j of L^-1 Rxx, then column j of C),
// then the DOMINANT eigenpair only (rank = 1 needs nothing else) by repeated squaring of C / tr C, which converges
// to v0 v0^H; d0 is the Rayleigh quotient q^H Rxx q of the back-substituted q = L^-H v0.
#pragma once
#include "common.h"
#include "k_cov.h"
#include "pk.h"

namespace disco {

#ifndef DISCO_SOLVE_PACKED
#define DISCO_SOLVE_PACKED 1
#endif
// Squaring stops with the square whose tau = tr(B^2) came within DISCO_SQUARING_DONE of 1: 1 - tau ~ 2 rho (rho = the
// sub-dominant weight (d1/d0)^(2^k) of the matrix that was squared), so the square that is kept carries rho^2 < 0.017.  The
// rest of the way is covered by DISCO_POWER_STEPS power steps v <- B v on the column picked from it: each costs 1/P of a
// squaring and multiplies the sub-dominant content by rho^2, leaving < 4 (0.017)^6 ~ 1e-10.  (Round 2 first ran the squaring
// itself down to 1 - tau < 1e-6: three more squarings for every wave, since a wave leaves with its slowest pencil.)
#ifndef DISCO_SQUARING_DONE
#define DISCO_SQUARING_DONE 0.2
#endif
#ifndef DISCO_POWER_STEPS
#define DISCO_POWER_STEPS 5
#endif
#ifndef DISCO_SQ_ROWS
#define DISCO_SQ_ROWS 1
#endif
#ifndef DISCO_SQUARINGS_MAX
#define DISCO_SQUARINGS_MAX 40
#endif
// A group of G <= 16 lanes never spans waves, so its LDS hand-offs need no s_barrier: a wave's DS instructions execute
// in issue order; what has to be prevented is the compiler moving a read above the write it depends on through another
// lane, and a read issuing before the wave's own writes were accepted (same fence as fft.h).
// The compiler-level memory clobber makes hipcc re-load LDS values after the fence instead of carrying them in registers
// from one phase to the next (it would keep the whole of L, P(P-1)/2 complex doubles, live between the two substitutions).
#ifndef DISCO_GROUP_SYNC
#define DISCO_GROUP_SYNC()                    \
    do {                                      \
        __builtin_amdgcn_s_waitcnt(0xC07F);   \
        __builtin_amdgcn_wave_barrier();      \
        asm volatile("" ::: "memory");        \
    } while (0)
#endif

constexpr double SOLVE_EPS = 2.220446049250313e-16;     // internal_formulas.py:6  sys.float_info.epsilon
constexpr double SOLVE_ETA = 1e6;                       // internal_formulas.py:7

// 1/sqrt(x) and 1/x in float64 from the hardware seeds (v_rsq_f64 / v_rcp_f64, ~26 good bits) plus two Newton steps:
// ~9 / ~5 instructions instead of the ~50 / ~25 of the IEEE-exact sqrt / divide expansions, 1-2 ulp; the inputs are
// O(1) covariance entries, far from the denormal / overflow corners the exact expansions exist for.
__device__ __forceinline__ double rsqrt64(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
__device__ __forceinline__ double rcp64(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    y = y * (2.0 - x * y);
    return y;
}

template <int P>
struct SolveGeom {
    static constexpr int G = P <= 4 ? 4 : (P <= 8 ? 8 : 16);
    static constexpr int THREADS = P <= 8 ? 128 : 64;           // keeps the two LDS matrices under 64 KiB
    static constexpr int PROBS = THREADS / G;
    // waves per SIMD the register allocation must leave room for (the LDS footprint of P > 8 would otherwise let hipcc
    // spread over all 512 registers, AGPR copies and scratch included)
    static constexpr int WPE = P <= 4 ? 6 : (P <= 7 ? 4 : (P == 8 ? 3 : 2));
    // LDS words (c64) per problem: L as a packed lower triangle (only that half is ever read), Y as a full matrix with a
    // padded row.  Packing L takes P = 7 from 1792 to 1344 bytes per problem -- the kernel's occupancy is LDS-bound.
    static constexpr int YW = (P % 8 == 0) ? P + 1 : P;     // row pitch of Y: padded only where P * 16 B would alias LDS banks
    // (+1 where that count is even: the per-problem stride in 16-byte words is then odd, so the 4 / 8 / 16 problems a wave serves
    // start on distinct bank groups and a broadcast read of "the same entry of every problem" is conflict-free)
    static constexpr int LSZ = (DISCO_SOLVE_PACKED ? P * (P + 1) / 2 : P * (P + 1)) | 1;
    __host__ __device__ static constexpr int lt(int i, int k) { return DISCO_SOLVE_PACKED ? i * (i + 1) / 2 + k : i * (P + 1) + k; }
};

// Where the pencils come from: either full row-major matrices (the disco_gevd_mwf_r1 ABI) or, inside the fused
// path, straight from the covariance kernels' chunk partials (upper triangle, (Rss, Rnn) interleaved per entry:
// part[g][chunk][f][q]), which saves materialising and re-reading the P x P matrices.
struct SolveSrc {
    const c32* Rss;
    const c32* Rnn;
    const float4* part;
    int F, chunks;
    float inv_T;
    // optional second source for the leading M_loc x M_loc block (step-1 partial sums re-used by step 2, see
    // k_step2_cov_fused<SKIPLOC>): entries (j, c) with max(j, c) < M_loc come from here.  M_loc = 0: unused.
    const float4* part_loc;
    int chunks_loc, M_loc;
};

// row j of both Hermitian matrices of problem pid: rs[c] = Rss[j][c], rn[c] = Rnn[j][c]
template <int P, bool FROM_PART>
__device__ __forceinline__ void solve_load_row(const SolveSrc& src, long long pid, int j, c32* rs, c32* rn) {
    if constexpr (!FROM_PART) {
#pragma unroll
        for (int c = 0; c < P; ++c) {
            rs[c] = src.Rss[pid * P * P + j * P + c];
            rn[c] = src.Rnn[pid * P * P + j * P + c];
        }
    } else {
        // the 64-bit part of every address once per lane (problem base in both sources, chunk stride), 32-bit triangle offsets per entry
        constexpr int NP = P * (P + 1) / 2;
        const long long g = pid / src.F;
        const int f = (int)(pid % src.F);
        const int ML = src.M_loc, NPL = ML * (ML + 1) / 2;
        const float4* pb = src.part + ((g * src.chunks) * src.F + f) * (long long)NP;
        const float4* pl = ML > 0 ? src.part_loc + ((g * src.chunks_loc) * src.F + f) * (long long)NPL : pb;
        const int cs = src.F * NP, csl = src.F * NPL;                // chunk strides (float4 units)
#pragma unroll
        for (int c = 0; c < P; ++c) {
            const bool up = c >= j;
            const int lo_ = up ? j : c, hi_ = up ? c : j;            // upper-triangle coordinates (lo_, hi_)
            const bool loc = hi_ < ML;
            const int q = lo_ * (loc ? ML : P) - (lo_ * (lo_ - 1)) / 2 + (hi_ - lo_);
            const float4* ptr = (loc ? pl : pb) + q;
            const int stride = loc ? csl : cs;
            const int nch = loc ? src.chunks_loc : src.chunks;
            // the chunk sums are combined in float64 and rounded ONCE: every chunk carries the rounding of its own run of frames only
            double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
            for (int ch = 0; ch < nch; ++ch) {
                const float4 v = ptr[(long long)ch * stride];
                sx += (double)v.x;
                sy += (double)v.y;
                sz += (double)v.z;
                sw += (double)v.w;
            }
            const double it = (double)src.inv_T, sg = up ? it : -it;     // lower triangle = conj(upper)
            rs[c] = make_float2((float)(sx * it), c == j ? 0.f : (float)(sy * sg));
            rn[c] = make_float2((float)(sz * it), c == j ? 0.f : (float)(sw * sg));
        }
    }
}

// ---- building blocks of the group solves (a group of G lanes, lane j owns row / column j; Lm, Ym: the group's LDS) ----------

// Cholesky factor of the Hermitian matrix whose row j lane j passes in `row` (only the lower triangle is used), left in Lm:
// strict lower triangle = L, diagonal slots = (1 / L[c][c], L[c][c]).
template <int P>
__device__ __forceinline__ int group_cholesky_factor(c64* Lm, const int j);

template <int P, class RowT>
__device__ __forceinline__ void group_cholesky(const RowT* row, c64* Lm, const int j) {
    using SG = SolveGeom<P>;
    if (j < P) {
#pragma unroll
        for (int c = 0; c < P; ++c) {
            if (c <= j) Lm[SG::lt(j, c)] = make_double2((double)row[c].x, (double)row[c].y);     // lower triangle only
        }
    }
    DISCO_GROUP_SYNC();
    (void)group_cholesky_factor<P>(Lm, j);
}

// the factorisation proper, on the lower triangle lane j's row was written to.  Returns how many of the first P - 1 pivots broke
// down (not positive): 0 <=> the leading (P - 1) x (P - 1) block is positive definite (Sylvester: what the tracking solver uses
// to know that its Rayleigh quotient lies above every other eigenvalue).
template <int P>
__device__ __forceinline__ int group_cholesky_factor(c64* Lm, const int j) {
    using SG = SolveGeom<P>;
    int broke = 0;

    // ---- Cholesky, one column per step; every lane keeps the (real) diagonal in registers.
    // Breakdown: the covariances arrive as float32, so a pivot below ~1e-7 of its diagonal entry is rounding noise (a
    // numerically singular Rnn: coherent noise at low frequencies, silent channels).  The reference's LAPACK pencil solver
    // returns finite numbers there (huge generalized eigenvalues, clamped to 1e6 by internal_formulas.py:59-60); here the
    // pivot is floored and the column below it zeroed, which bounds every later quantity instead of amplifying noise by
    // 1/sqrt(pivot) per breakdown.  Well-conditioned pencils never take this branch.
    // The diagonal slot of column c (Rnn[c][c], dead once step c has read it) then parks (1 / L[c][c], L[c][c]) for the
    // substitutions below and at the very end: no per-column array stays in registers.  It is written one step late, by
    // lane 0, after the fence that ends the step in which every lane read the old content.
    double rd_prev = 0.0, d_prev = 0.0;
#pragma unroll
    for (int c = 0; c < P; ++c) {
        DISCO_SCHED_FENCE();
        if (c > 0 && j == 0) Lm[SG::lt(c - 1, c - 1)] = make_double2(rd_prev, d_prev);
        const double a_cc = Lm[SG::lt(c, c)].x;
        double d2 = a_cc;
#pragma unroll
        for (int k = 0; k < c; ++k) d2 -= Lm[SG::lt(c, k)].x * Lm[SG::lt(c, k)].x + Lm[SG::lt(c, k)].y * Lm[SG::lt(c, k)].y;
        const double fl = fmax(1e-7 * a_cc, 1e-30);
        const bool brk = !(d2 >= fl);                   // also true for NaN
        if (c < P - 1 && brk) ++broke;
        const double d2c = brk ? fl : d2;
        const double rd = rsqrt64(d2c);                 // 1 / L[c][c]
        rd_prev = rd;
        d_prev = d2c * rd;                              //     L[c][c]
        if (j > c && j < P) {
            c64 s = Lm[SG::lt(j, c)];
#pragma unroll
            for (int k = 0; k < c; ++k) s = zsub(s, zmulc(Lm[SG::lt(j, k)], Lm[SG::lt(c, k)]));
            Lm[SG::lt(j, c)] = brk ? make_double2(0.0, 0.0) : zscale(s, rd);
        }
        DISCO_GROUP_SYNC();
    }
    if (j == 0) Lm[SG::lt(P - 1, P - 1)] = make_double2(rd_prev, d_prev);
    DISCO_GROUP_SYNC();
    return broke;
}

// Dominant eigenvector of the Hermitian matrix whose column j lane j passes in g (destroyed): v0 (unit norm, every lane gets
// all of it).  Returns false when the matrix is zero or not finite (v0 = e0 then).
template <int P>
__device__ __forceinline__ bool group_dominant(c64* g, c64 (*Ym)[SolveGeom<P>::YW], const int j, c64* v0) {
    constexpr int G = SolveGeom<P>::G;
    bool done;
    {
        double trl = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (i == j) trl = g[i].x;
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) trl += __shfl_xor(trl, off, G);
        const bool ok = trl > 0.0 && trl < 1.7e308;            // false for NaN / inf / the zero matrix (Rxx = 0)
        const double rt = ok ? rcp64(trl) : 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) g[i] = ok ? zscale(g[i], rt) : make_double2(0.0, 0.0);
        done = !ok;
    }
    for (int it = 0; it < DISCO_SQUARINGS_MAX; ++it) {
        DISCO_GROUP_SYNC();                                    // previous readers of Ym (Y above / the last square) are done
        if (j < P) {
#pragma unroll
            for (int i = 0; i < P; ++i) Ym[i][j] = g[i];
        }
        DISCO_GROUP_SYNC();
        c64 nn[P];
        c64 tc = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            // DISCO_SQ_ROWS rows of B at a time: without the fences hipcc hoists all P^2 LDS loads (4 registers each) above
            // the first multiply and the kernel drops to one wave per SIMD with spills at P = 15
            if (i % DISCO_SQ_ROWS == 0) DISCO_SCHED_FENCE();
            c64 a = make_double2(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const c64 b = Ym[i][k];
                a.x = fma(b.x, g[k].x, fma(-b.y, g[k].y, a.x));
                a.y = fma(b.x, g[k].y, fma(b.y, g[k].x, a.y));
            }
            nn[i] = a;
            if (i == j) tc = a;
        }
        DISCO_SCHED_FENCE();
        // tau = tr(B^2) as a COMPLEX number.  For an exactly Hermitian B it is real, but rounding (and inputs whose two
        // triangles were accumulated separately, as the online kernel's) leave B = (1 + i eps) x a Hermitian matrix, and a
        // normaliser that ignores Im tau lets that phase DOUBLE with every squaring (B^2 carries (1 + i eps)^2) until tau
        // turns negative -- measured on the MI355X: 1 - Re tau grew 4x per iteration from 1e-11.  Dividing by the complex
        // trace removes the common complex scale altogether; the remaining perturbations of the fixed point v0 v0^H are
        // either annihilated or carried unchanged, so squaring on after convergence is harmless.
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) {
            tc.x += __shfl_xor(tc.x, off, G);
            tc.y += __shfl_xor(tc.y, off, G);
        }
        const double den = tc.x * tc.x + tc.y * tc.y;
        const double rden = den > 0.0 ? rcp64(den) : 0.0;
        const c64 itau = make_double2(tc.x * rden, -tc.y * rden);
        if (!done) {                                           // a finished group idles until the slowest of its wave is through
#pragma unroll
            for (int i = 0; i < P; ++i) g[i] = zmul(nn[i], itau);
        }
        done = done || (1.0 - tc.x < DISCO_SQUARING_DONE) || !(den > 0.0);
        if (!__any(!done)) break;                              // wave-uniform exit                              // wave-uniform exit: finished groups keep squaring a projector
    }

    // L is read again by the back substitution below: make the compiler re-load it from LDS there instead of carrying
    // the whole strict lower triangle (P(P-1)/2 complex doubles, 84 VGPRs at P = 7) in registers across the squaring loop
    asm volatile("" ::: "memory");

    // ---- B = v0 v0^H: every non-zero column is a multiple of v0; take the longest (|v0[j]| largest; ties: lowest lane)
    double nrm = 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) nrm += g[i].x * g[i].x + g[i].y * g[i].y;
    double best = nrm;
    int bj = j;
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) {
        const double ob = __shfl_xor(best, off, G);
        const int oj = __shfl_xor(bj, off, G);
        if (ob > best || (ob == best && oj < bj)) {
            best = ob;
            bj = oj;
        }
    }
    const bool have = best > 0.0;                          // false: C = 0 or not finite -> v0 = e0 (d0 is clamped below)
    const double rb = have ? rsqrt64(best) : 0.0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        v0[i].x = __shfl(g[i].x, bj, G);
        v0[i].y = __shfl(g[i].y, bj, G);
        if (have) v0[i] = zscale(v0[i], rb);
        else v0[i] = make_double2(i == 0 ? 1.0 : 0.0, 0.0);
    }
    // ---- power steps on the kept square: (B v)[j] = sum_i conj(B[i][j]) v[i] (B is Hermitian), lane j from its own column;
    // the new vector goes round the group through row 0 of Ym.  The top eigenvalue of the trace-normalised B is >= 1/P, so
    // the length can wait until the end.
#pragma unroll 1
    for (int s = 0; s < DISCO_POWER_STEPS; ++s) {
        c64 u = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            u.x = fma(g[i].x, v0[i].x, fma(g[i].y, v0[i].y, u.x));
            u.y = fma(g[i].x, v0[i].y, fma(-g[i].y, v0[i].x, u.y));
        }
        DISCO_GROUP_SYNC();
        if (j < P) Ym[0][j] = u;
        DISCO_GROUP_SYNC();
        if (have) {
#pragma unroll
            for (int i = 0; i < P; ++i) v0[i] = Ym[0][i];
        }
    }
    if (DISCO_POWER_STEPS > 0 && have) {
        double n2 = 0.0;
#pragma unroll
        for (int i = 0; i < P; ++i) n2 = fma(v0[i].x, v0[i].x, fma(v0[i].y, v0[i].y, n2));
        const double rn = rsqrt64(n2);
#pragma unroll
        for (int i = 0; i < P; ++i) v0[i] = zscale(v0[i], rn);
    }
    return have;
}

// q = L^-H v (back substitution against the factor group_cholesky left in Lm); every lane computes all of q
template <int P>
__device__ __forceinline__ void group_back_substitute(const c64* v0, const c64* Lm, c64* q) {
    using SG = SolveGeom<P>;
    // ---- q = L^-H v0 (back substitution); the diagonal slots hold (1 / L[i][i], L[i][i])
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
        DISCO_SCHED_FENCE();
        c64 a = v0[i];
#pragma unroll
        for (int k = i + 1; k < P; ++k) a = zsub(a, zmul(make_double2(Lm[SG::lt(k, i)].x, -Lm[SG::lt(k, i)].y), q[k]));
        q[i] = zscale(a, Lm[SG::lt(i, i)].x);
        // anchors the row's arithmetic between the fences: otherwise every column of L is loaded (and spilled) before the first multiply
        DISCO_CONSUME(q[i].x);
        DISCO_CONSUME(q[i].y);
    }
}

// x = L^-1 b (forward substitution), in place; every lane computes all of it
template <int P>
__device__ __forceinline__ void group_forward_substitute(c64* b, const c64* Lm) {
    using SG = SolveGeom<P>;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        DISCO_SCHED_FENCE();
        c64 a = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) a = zsub(a, zmul(Lm[SG::lt(i, k)], b[k]));
        b[i] = zscale(a, Lm[SG::lt(i, i)].x);
    }
}

// The solve proper for the group of G lanes that owns one pencil: lane j (< P) passes row j of Rxx (rowA) and of Rnn
// (rowB); Lm / Ym are the group's two LDS matrices.  Returns this lane's component of t1 and the scalar gain
// d0 / (d0 + mu)  (w_j = t1_j * gain).  Contains wave-level fences and wave-wide votes: every lane of a wave must call it,
// the same number of times.  REENTER: a fence first, so that a previous call's readers of Lm / Ym are done (callers in a loop).
template <int P, bool REENTER>
__device__ __forceinline__ void gevd_solve_group(const c32* rowA, const c32* rowB, c64* Lm, c64 (*Ym)[SolveGeom<P>::YW],
                                                 const int j, const double mu, c64& t1_j, double& gain_out) {
    constexpr int G = SolveGeom<P>::G;
    using SG = SolveGeom<P>;
    if constexpr (REENTER) DISCO_GROUP_SYNC();
    group_cholesky<P>(rowB, Lm, j);

    // ---- column j of Y = L^-1 Rxx   (Rxx[i][j] = conj(Rxx[j][i]): read row j, contiguous)
    c64 y[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        DISCO_SCHED_FENCE();                            // one row of L at a time (see the squaring loop)
        c64 a = make_double2((double)rowA[i].x, -(double)rowA[i].y);
#pragma unroll
        for (int k = 0; k < i; ++k) a = zsub(a, zmul(Lm[SG::lt(i, k)], y[k]));
        y[i] = zscale(a, Lm[SG::lt(i, i)].x);
        // y is only stored under `j < P` below: without a use here hipcc sinks the whole substitution into that branch
        // while its LDS loads stay outside, i.e. all of L is loaded (and spilled) first
        DISCO_CONSUME(y[i].x);
        DISCO_CONSUME(y[i].y);
    }
    if (j < P) {
#pragma unroll
        for (int i = 0; i < P; ++i) Ym[i][j] = y[i];
    }
    DISCO_GROUP_SYNC();

    // ---- column j of C = L^-1 Y^H  (C Hermitian): rhs = conj(row j of Y)
    c64 g[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        DISCO_SCHED_FENCE();
        c64 a = make_double2(0.0, 0.0);
        if (j < P) a = make_double2(Ym[j][i].x, -Ym[j][i].y);
#pragma unroll
        for (int k = 0; k < i; ++k) a = zsub(a, zmul(Lm[SG::lt(i, k)], g[k]));
        g[i] = zscale(a, Lm[SG::lt(i, i)].x);
        DISCO_CONSUME(g[i].x);                          // as above: lanes >= P carry zero columns, a branch hipcc would sink into
        DISCO_CONSUME(g[i].y);
    }

    asm volatile("" ::: "memory");                           // L is not needed again before the back substitution

    // ---- dominant eigenpair of C by repeated squaring.  rank = 1 keeps only (d0, v0) (internal_formulas.py:63-69), so
    // the full diagonalisation a Jacobi solver performs (what this kernel did before: 5-7 sweeps of P(P-1)/2 rotations)
    // is not needed: with B_0 = C / tr C,  B_{k+1} = B_k^2 / tr(B_k^2)  converges to v0 v0^H and the sub-dominant
    // directions decay like (d1/d0)^(2^k) -- 3-6 squarings for the ratios 0.5-0.9 met on real covariances, P^2 complex
    // multiply-adds per lane each (a Jacobi SWEEP costs ~5 P^2).  tau_k = tr(B_k^2) = ||B_k||_F^2 <= 1 doubles as the
    // normaliser and the convergence measure: 1 - tau ~ 2 (d1/d0)^(2^k); once it is below DISCO_SQUARING_DONE a few power
    // steps on the square just formed finish the job at 1/P of the price of a squaring each.  Lane j owns column j; the columns meet through the group's LDS matrix Ym (wave-level
    // fences only: a group never spans waves).  An exactly repeated top eigenvalue never converges (tau -> 1/m) and
    // stops at the iteration cap with a vector of the dominant subspace, which is all any solver can return there.
    // The loop is wave-uniform (vote on the exit): groups that are done keep their B and idle.
    c64 v0[P];
    const bool have = group_dominant<P>(g, Ym, j, v0);

    c64 q[P];
    group_back_substitute<P>(v0, Lm, q);
    const double l00 = Lm[SG::lt(0, 0)].y;
    // ---- d0 = v0^H C v0 = q^H Rxx q  (q^H Rnn q = |v0|^2 = 1): lane j forms (Rxx q)_j from its row of Rxx
    double d0;
    {
        c64 sj = make_double2(0.0, 0.0), qj = make_double2(0.0, 0.0);
#pragma unroll
        for (int c = 0; c < P; ++c) {
            sj.x = fma((double)rowA[c].x, q[c].x, fma(-(double)rowA[c].y, q[c].y, sj.x));
            sj.y = fma((double)rowA[c].x, q[c].y, fma((double)rowA[c].y, q[c].x, sj.y));
            if (c == j) qj = q[c];
        }
        double e = j < P ? qj.x * sj.x + qj.y * sj.y : 0.0;      // Re(conj(q_j) (Rxx q)_j)
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) e += __shfl_xor(e, off, G);
        d0 = have ? e : 0.0;
    }
    const double dcl = fmin(fmax(d0, SOLVE_EPS), SOLVE_ETA);
    const c64 gsc = make_double2(l00 * v0[0].x, -l00 * v0[0].y);     // L[0,0] conj(v0[0]) = (Q^-1)[0,0]
    const double gain = dcl / (dcl + mu);
    t1_j = make_double2(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        if (i == j) t1_j = zmul(q[i], gsc);
    }
    gain_out = gain;
}

// (A mixed-precision form of this solve -- float32 squarings on packed instructions and a float64 Rayleigh-quotient finish, option
// "solve_f32" of rounds 2-4 -- measured slower than the all-float64 route, 1.26 against 1.07 ms per C3 launch at P = 7 and 16.2 against
// 14.4 ms per C5 step at P = 15, and was removed in round 5: profiles/r03_design_and_experiment_log.md, git history.)

template <int P, bool FROM_PART>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(SolveGeom<P>::THREADS, SolveGeom<P>::WPE) void k_gevd_mwf_r1(SolveSrc src, long long n_prob, double mu,
                                                                        c32* __restrict__ w_out, c32* __restrict__ t1_out) {
    constexpr int G = SolveGeom<P>::G, PROBS = SolveGeom<P>::PROBS;
    __shared__ c64 s_L[PROBS][SolveGeom<P>::LSZ];       // lower triangle: L (strict) ; diagonal keeps Rnn[c][c]
    __shared__ c64 s_Y[PROBS][P * SolveGeom<P>::YW];      // second region: the float64 matrix Y
    const int j = threadIdx.x % G;             // column owned by this lane
    const int slot = threadIdx.x / G;
    const long long pid = (long long)blockIdx.x * PROBS + slot;
    const bool live = pid < n_prob;
    const bool col = live && j < P;

    // ---- row j of both matrices
    c32 rowA[P], rowB[P];
    if (col) {
        solve_load_row<P, FROM_PART>(src, pid, j, rowA, rowB);
    } else {
#pragma unroll
        for (int c = 0; c < P; ++c) {
            rowA[c] = make_float2(0.f, 0.f);
            rowB[c] = make_float2(c == j ? 1.f : 0.f, 0.f);
        }
    }
    c64 t1;
    double gain;
    gevd_solve_group<P, false>(rowA, rowB, s_L[slot], reinterpret_cast<c64(*)[SolveGeom<P>::YW]>(s_Y[slot]), j, mu, t1, gain);
    if (col) {
        if (t1_out) t1_out[pid * P + j] = make_float2((float)t1.x, (float)t1.y);
        w_out[pid * P + j] = make_float2((float)(t1.x * gain), (float)(t1.y * gain));
    }
}

// ---- the other two branches of intern_filter (internal_formulas.py:45-54, 74-76): dead on the hot path, offered so that the
// function's whole surface (its DEFAULT type is 'r1-mwf') is there.  Same mapping and the same group primitives:
//   'mwf'    : Wint = ((Rnn + Rxx)^-1 Rxx)[:, 0]                               -- one Cholesky solve
//   'r1-mwf' : Rxx1 = |Dmax| x x^H (dominant eigenpair of Rxx alone); P = Rnn^-1 Rxx1; Wint = P[:, 0] / (mu + tr P)
//              = |Dmax| conj(x_0) u / (mu + |Dmax| x^H u),  u = Rnn^-1 x      -- squaring + one Cholesky solve
// (np.linalg.lstsq's minimum-norm answer for a singular matrix is NOT reproduced: the pivot floor of group_cholesky applies.)
constexpr int FILTER_R1_MWF = 1, FILTER_MWF = 2;

template <int P, int TYPE>
__global__ __launch_bounds__(SolveGeom<P>::THREADS, SolveGeom<P>::WPE) void k_mwf_variants(const c32* __restrict__ Rxx,
                                                                                          const c32* __restrict__ Rnn, long long n_prob,
                                                                                          double mu, c32* __restrict__ w_out) {
    constexpr int G = SolveGeom<P>::G, PROBS = SolveGeom<P>::PROBS;
    using SG = SolveGeom<P>;
    __shared__ c64 s_L[PROBS][SolveGeom<P>::LSZ];
    __shared__ c64 s_Y[PROBS][P][SolveGeom<P>::YW];
    const int j = threadIdx.x % G, slot = threadIdx.x / G;
    const long long pid = (long long)blockIdx.x * PROBS + slot;
    const bool col = pid < n_prob && j < P;
    c32 rowA[P], rowB[P];                       // row j of Rxx / of the matrix to factor
#pragma unroll
    for (int c = 0; c < P; ++c) {
        rowA[c] = col ? Rxx[pid * P * P + j * P + c] : make_float2(0.f, 0.f);
        const c32 b = col ? Rnn[pid * P * P + j * P + c] : make_float2(c == j ? 1.f : 0.f, 0.f);
        rowB[c] = TYPE == FILTER_MWF ? make_float2(b.x + rowA[c].x, b.y + rowA[c].y) : b;
    }
    c64* Lm = s_L[slot];
    c64 rhs[P];                                 // the right-hand side, all of it in every lane
    double scale = 1.0;                         // 'r1-mwf': |Dmax|
    if constexpr (TYPE == FILTER_MWF) {         // column 0 of Rxx = conj(row 0), which lane 0 of the group holds
#pragma unroll
        for (int i = 0; i < P; ++i)
            rhs[i] = make_double2((double)__shfl(rowA[i].x, 0, G), -(double)__shfl(rowA[i].y, 0, G));
    } else {                                    // dominant eigenpair of Rxx: column j = conj(row j)
        c64 g[P];
#pragma unroll
        for (int i = 0; i < P; ++i) g[i] = make_double2((double)rowA[i].x, -(double)rowA[i].y);
        group_dominant<P>(g, s_Y[slot], j, rhs);
        // Dmax = x^H Rxx x: lane j forms (Rxx x)_j from its row
        c64 sj = make_double2(0.0, 0.0), xj = make_double2(0.0, 0.0);
#pragma unroll
        for (int c = 0; c < P; ++c) {
            sj.x = fma((double)rowA[c].x, rhs[c].x, fma(-(double)rowA[c].y, rhs[c].y, sj.x));
            sj.y = fma((double)rowA[c].x, rhs[c].y, fma((double)rowA[c].y, rhs[c].x, sj.y));
            if (c == j) xj = rhs[c];
        }
        double e = j < P ? xj.x * sj.x + xj.y * sj.y : 0.0;
#pragma unroll
        for (int off = G / 2; off >= 1; off >>= 1) e += __shfl_xor(e, off, G);
        scale = fabs(e);
    }
    group_cholesky<P>(rowB, Lm, j);
    c64 x0 = rhs[0];                            // 'r1-mwf' needs x_0 and x^H u after the solve has overwritten nothing: keep x
    c64 xk[TYPE == FILTER_R1_MWF ? P : 1];
    if constexpr (TYPE == FILTER_R1_MWF) {
#pragma unroll
        for (int i = 0; i < P; ++i) xk[i] = rhs[i];
    }
    group_forward_substitute<P>(rhs, Lm);
    c64 u[P];
    group_back_substitute<P>(rhs, Lm, u);
    c64 wj = make_double2(0.0, 0.0);
    if constexpr (TYPE == FILTER_MWF) {
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (i == j) wj = u[i];
    } else {
        c64 xu = make_double2(0.0, 0.0);        // x^H u
#pragma unroll
        for (int i = 0; i < P; ++i) xu = zadd(xu, zmul(make_double2(xk[i].x, -xk[i].y), u[i]));
        const c64 den = make_double2(mu + scale * xu.x, scale * xu.y);
        const double rd = rcp64(den.x * den.x + den.y * den.y);
        const c64 num = zscale(make_double2(x0.x, -x0.y), scale);                   // |Dmax| conj(x_0)
        const c64 f = zmul(num, make_double2(den.x * rd, -den.y * rd));             // / (mu + tr P)
#pragma unroll
        for (int i = 0; i < P; ++i)
            if (i == j) wj = zmul(u[i], f);
    }
    if (col) w_out[pid * P + j] = make_float2((float)wj.x, (float)wj.y);
}

}  // namespace disco
