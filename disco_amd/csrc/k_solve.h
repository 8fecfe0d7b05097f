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
// Mapping: a group of G = 4 / 8 / 16 lanes owns one problem, lane j owns column j.  Everything is float64:
// cooperative Cholesky through LDS, two forward substitutions (column j of L^-1 Rxx, then column j of C),
// then a ONE-SIDED (Hestenes) Jacobi on the columns of C with a round-robin tournament: each round every lane
// fetches its partner's column with shuffles, both compute the same plane rotation, each updates its own
// column.  On convergence the columns are d_j v_j; the longest one gives (d0, v0).
#pragma once
#include "common.h"
#include "k_cov.h"

namespace disco {

#ifndef DISCO_SOLVE_PACKED
#define DISCO_SOLVE_PACKED 1
#endif
// a sweep whose every pair was orthogonal to sqrt(DISCO_JACOBI_DONE) before being rotated ends the iteration
#ifndef DISCO_JACOBI_DONE
#define DISCO_JACOBI_DONE 1e-14
#endif

constexpr double SOLVE_EPS = 2.220446049250313e-16;     // internal_formulas.py:6  sys.float_info.epsilon
constexpr double SOLVE_ETA = 1e6;                       // internal_formulas.py:7

// 1/sqrt(x) and 1/x in float64 from the hardware seeds (v_rsq_f64 / v_rcp_f64, ~26 good bits) plus two Newton steps:
// ~9 / ~5 instructions instead of the ~50 / ~25 of the IEEE-exact sqrt / divide expansions, 1-2 ulp.  The Jacobi
// rotations only need cs^2 + sn^2 = 1 and |phase| = 1 to rounding, which these deliver; the inputs are O(1)
// covariance entries, far from the denormal / overflow corners the exact expansions exist for.
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
    // LDS words (c64) per problem: L as a packed lower triangle (only that half is ever read), Y as a full matrix with a
    // padded row.  Packing L takes P = 7 from 1792 to 1344 bytes per problem -- the kernel's occupancy is LDS-bound.
    static constexpr int YW = (P % 8 == 0) ? P + 1 : P;     // row pitch of Y: padded only where P * 16 B would alias LDS banks
    static constexpr int LSZ = DISCO_SOLVE_PACKED ? P * (P + 1) / 2 : P * (P + 1);
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
        constexpr int NP = P * (P + 1) / 2;
        const long long g = pid / src.F;
        const int f = (int)(pid % src.F);
#pragma unroll
        for (int c = 0; c < P; ++c) {
            const bool up = c >= j;
            const int lo_ = up ? j : c, hi_ = up ? c : j;            // upper-triangle coordinates (lo_, hi_)
            const bool loc = hi_ < src.M_loc;
            const int Pq = loc ? src.M_loc : P;
            const int q = lo_ * Pq - (lo_ * (lo_ - 1)) / 2 + (hi_ - lo_);
            const float4* base = loc ? src.part_loc : src.part;
            const int nch = loc ? src.chunks_loc : src.chunks;
            const long long npq = loc ? (long long)(src.M_loc * (src.M_loc + 1) / 2) : (long long)NP;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int ch = 0; ch < nch; ++ch) {
                const float4 v = base[(((g * nch + ch) * src.F) + f) * npq + q];
                s.x += v.x;
                s.y += v.y;
                s.z += v.z;
                s.w += v.w;
            }
            const float sg = up ? src.inv_T : -src.inv_T;           // lower triangle = conj(upper)
            rs[c] = make_float2(s.x * src.inv_T, c == j ? 0.f : s.y * sg);
            rn[c] = make_float2(s.z * src.inv_T, c == j ? 0.f : s.w * sg);
        }
    }
}

// The solve proper for the group of G lanes that owns one pencil: lane j (< P) passes row j of Rxx (rowA) and of Rnn
// (rowB); Lm / Ym are the group's two LDS matrices.  Returns this lane's component of t1 and the scalar gain
// d0 / (d0 + mu)  (w_j = t1_j * gain).  Contains block-level barriers: every thread of the block must call it, the same
// number of times.  REENTER: a barrier first, so that a previous call's readers of Lm / Ym are done (callers in a loop).
template <int P, bool REENTER>
__device__ __forceinline__ void gevd_solve_group(const c32* rowA, const c32* rowB, c64* Lm, c64 (*Ym)[SolveGeom<P>::YW],
                                                 const int j, const double mu, c64& t1_j, double& gain_out) {
    constexpr int G = SolveGeom<P>::G;
    using SG = SolveGeom<P>;
    if constexpr (REENTER) __syncthreads();
    if (j < P) {
#pragma unroll
        for (int c = 0; c < P; ++c) {
            if (c <= j) Lm[SG::lt(j, c)] = make_double2((double)rowB[c].x, (double)rowB[c].y);     // lower triangle only
        }
    }
    __syncthreads();

    // ---- Cholesky, one column per step; every lane keeps the (real) diagonal in registers.
    // Breakdown: the covariances arrive as float32, so a pivot below ~1e-7 of its diagonal entry is rounding noise (a
    // numerically singular Rnn: coherent noise at low frequencies, silent channels).  The reference's LAPACK pencil solver
    // returns finite numbers there (huge generalized eigenvalues, clamped to 1e6 by internal_formulas.py:59-60); here the
    // pivot is floored and the column below it zeroed, which bounds every later quantity instead of amplifying noise by
    // 1/sqrt(pivot) per breakdown.  Well-conditioned pencils never take this branch.
    double dd[P], rdd[P];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        const double a_cc = Lm[SG::lt(c, c)].x;
        double d2 = a_cc;
#pragma unroll
        for (int k = 0; k < c; ++k) d2 -= Lm[SG::lt(c, k)].x * Lm[SG::lt(c, k)].x + Lm[SG::lt(c, k)].y * Lm[SG::lt(c, k)].y;
        const double fl = fmax(1e-7 * a_cc, 1e-30);
        const bool brk = !(d2 >= fl);                   // also true for NaN
        const double d2c = brk ? fl : d2;
        const double rd = rsqrt64(d2c);                 // 1 / L[c][c]
        dd[c] = d2c * rd;                               //     L[c][c]
        rdd[c] = rd;
        if (j > c && j < P) {
            c64 s = Lm[SG::lt(j, c)];
#pragma unroll
            for (int k = 0; k < c; ++k) s = zsub(s, zmulc(Lm[SG::lt(j, k)], Lm[SG::lt(c, k)]));
            Lm[SG::lt(j, c)] = brk ? make_double2(0.0, 0.0) : zscale(s, rd);
        }
        __syncthreads();
    }

    // ---- column j of Y = L^-1 Rxx   (Rxx[i][j] = conj(Rxx[j][i]): read row j, contiguous)
    c64 y[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        c64 a = make_double2((double)rowA[i].x, -(double)rowA[i].y);
#pragma unroll
        for (int k = 0; k < i; ++k) a = zsub(a, zmul(Lm[SG::lt(i, k)], y[k]));
        y[i] = zscale(a, rdd[i]);
    }
    if (j < P) {
#pragma unroll
        for (int i = 0; i < P; ++i) Ym[i][j] = y[i];
    }
    __syncthreads();

    // ---- column j of C = L^-1 Y^H  (C Hermitian): rhs = conj(row j of Y)
    c64 g[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        c64 a = make_double2(0.0, 0.0);
        if (j < P) a = make_double2(Ym[j][i].x, -Ym[j][i].y);
#pragma unroll
        for (int k = 0; k < i; ++k) a = zsub(a, zmul(Lm[SG::lt(i, k)], g[k]));
        g[i] = zscale(a, rdd[i]);
    }

    // ---- one-sided Jacobi, round-robin over G players (lanes >= P carry zero columns)
    // Convergence is quadratic: once every pair of a sweep was already orthogonal to 1e-7 (g2 <= 1e-14 alpha beta) before
    // being rotated, the sweep leaves it at ~1e-14 and a further (verification) sweep would rotate nothing that matters.
    for (int sweep = 0; sweep < 40; ++sweep) {
        int rotated = 0;
        for (int rd = 0; rd < G - 1; ++rd) {
            int pj;
            if (j == G - 1) pj = rd;
            else if (j == rd) pj = G - 1;
            else {
                pj = 2 * rd - j;
                pj = pj < 0 ? pj + (G - 1) : (pj >= G - 1 ? pj - (G - 1) : pj);
            }
            c64 o[P];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                o[i].x = __shfl(g[i].x, pj, G);
                o[i].y = __shfl(g[i].y, pj, G);
            }
            const bool lo = j < pj;                       // own column plays "p" (first), partner's plays "q"
            // |own|^2, |other|^2 and own^H other, then ordered as (alpha, beta, gamma) = (|gp|^2, |gq|^2, gp^H gq):
            // four selects on scalars instead of 2P selects on the columns
            double n_own = 0.0, n_oth = 0.0, cr = 0.0, ci = 0.0;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                n_own += g[i].x * g[i].x + g[i].y * g[i].y;
                n_oth += o[i].x * o[i].x + o[i].y * o[i].y;
                cr += g[i].x * o[i].x + g[i].y * o[i].y;
                ci += g[i].x * o[i].y - g[i].y * o[i].x;
            }
            const double alpha = lo ? n_own : n_oth, beta = lo ? n_oth : n_own;
            const double gr = cr, gi = lo ? ci : -ci;     // gq^H gp = conj(gp^H gq)
            const double g2 = gr * gr + gi * gi;
            if (g2 > 1e-28 * alpha * beta && g2 > 0.0) {
                if (g2 > DISCO_JACOBI_DONE * alpha * beta) rotated = 1;
                const double rg = rsqrt64(g2);                                  // 1 / |gamma|
                const double zeta = 0.5 * (beta - alpha) * rg;
                const double hz = 1.0 + zeta * zeta;
                double t = rcp64(fabs(zeta) + hz * rsqrt64(hz));
                t = zeta >= 0.0 ? t : -t;
                const double cs = rsqrt64(1.0 + t * t), sn = cs * t;
                const c64 ph = make_double2(gr * rg, gi * rg);                  // e^{i phi}
                // gp' = cs gp - sn e^{-i phi} gq ;  gq' = sn e^{i phi} gp + cs gq
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    if (lo) {
                        const c64 r = zmulc(o[i], ph);                // e^{-i phi} gq
                        g[i] = make_double2(cs * g[i].x - sn * r.x, cs * g[i].y - sn * r.y);
                    } else {
                        const c64 r = zmul(o[i], ph);                 // e^{i phi} gp
                        g[i] = make_double2(sn * r.x + cs * g[i].x, sn * r.y + cs * g[i].y);
                    }
                }
            }
        }
        if (!__any(rotated)) break;
    }

    // L is read again by the back substitution below: make the compiler re-load it from LDS there instead of carrying
    // the whole strict lower triangle (P(P-1)/2 complex doubles, 84 VGPRs at P = 7) in registers across the Jacobi loop
    asm volatile("" ::: "memory");

    // ---- longest column = d0 v0 ; arg-max over the group (ties: lowest lane)
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
    c64 v0[P];
    const double rb = best > 0.0 ? rsqrt64(best) : 0.0;
    const double d0 = best * rb;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        v0[i].x = __shfl(g[i].x, bj, G);
        v0[i].y = __shfl(g[i].y, bj, G);
        if (d0 > 0.0) v0[i] = zscale(v0[i], rb);
        else v0[i] = make_double2(i == 0 ? 1.0 : 0.0, 0.0);
    }

    // ---- q = L^-H v0 (back substitution), then scale
    c64 q[P];
#pragma unroll
    for (int i = P - 1; i >= 0; --i) {
        c64 a = v0[i];
#pragma unroll
        for (int k = i + 1; k < P; ++k) a = zsub(a, zmul(make_double2(Lm[SG::lt(k, i)].x, -Lm[SG::lt(k, i)].y), q[k]));
        q[i] = zscale(a, rdd[i]);
    }
    const double dcl = fmin(fmax(d0, SOLVE_EPS), SOLVE_ETA);
    const c64 gsc = make_double2(dd[0] * v0[0].x, -dd[0] * v0[0].y);     // L[0,0] conj(v0[0]) = (Q^-1)[0,0]
    const double gain = dcl / (dcl + mu);
    t1_j = make_double2(0.0, 0.0);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        if (i == j) t1_j = zmul(q[i], gsc);
    }
    gain_out = gain;
}

template <int P, bool FROM_PART>
__global__ __launch_bounds__(SolveGeom<P>::THREADS) void k_gevd_mwf_r1(SolveSrc src, long long n_prob, double mu,
                                                                        c32* __restrict__ w_out, c32* __restrict__ t1_out) {
    constexpr int G = SolveGeom<P>::G, PROBS = SolveGeom<P>::PROBS;
    __shared__ c64 s_L[PROBS][SolveGeom<P>::LSZ];       // lower triangle: L (strict) ; diagonal keeps Rnn[c][c]
    __shared__ c64 s_Y[PROBS][P][SolveGeom<P>::YW];
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
    gevd_solve_group<P, false>(rowA, rowB, s_L[slot], s_Y[slot], j, mu, t1, gain);
    if (col) {
        if (t1_out) t1_out[pid * P + j] = make_float2((float)t1.x, (float)t1.y);
        w_out[pid * P + j] = make_float2((float)(t1.x * gain), (float)(t1.y * gain));
    }
}

}  // namespace disco
