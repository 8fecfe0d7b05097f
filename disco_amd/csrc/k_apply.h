// Filter-and-sum (tango.py:369-376, 445-450): out[t,f] = sum_p c(w[f,p]) v[p,t,f] with
// v = [X_k ; Z_j (j<k) ; Z_j (j>k)].  Flat over (t, f): consecutive lanes take consecutive bins, so X, Z and
// out are streamed fully coalesced; the (F x P) filter of the node stays in L1/L2.
#pragma once
#include "common.h"

namespace disco {

template <int M, int KR>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(256) void k_apply(const c32* __restrict__ X, const c32* __restrict__ Z,
                                                const c32* __restrict__ w, c32* __restrict__ out,
                                                int K, int T, int F, int conj_w, int blocks_per_node, int Kl, int k0, int zblk,
                                                long long R) {
    constexpr int P = M + KR;
    const long long g = blockIdx.x / blocks_per_node;            // local unit r*Kl + kl (see CovArgs)
    const int b = (int)(blockIdx.x % blocks_per_node);
    const long long r = g / Kl;
    const int k = k0 + (int)(g % Kl);
    const long long TF = (long long)T * F;
    const float sgn = conj_w ? -1.f : 1.f;
    const c32* wg = w + g * F * (long long)P;
    for (long long tf = (long long)b * blockDim.x + threadIdx.x; tf < TF; tf += (long long)blocks_per_node * blockDim.x) {
        const int f = (int)(tf % F);
        const c32* wf = wg + f * P;
        const c32* xp = X + (g * TF + tf) * M;
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const c32 x = xp[i];
            const c32 ww = make_float2(wf[i].x, sgn * wf[i].y);
            ar = fmaf(ww.x, x.x, fmaf(-ww.y, x.y, ar));
            ai = fmaf(ww.x, x.y, fmaf(ww.y, x.x, ai));
        }
#pragma unroll
        for (int jj = 0; jj < KR; ++jj) {
            const int j = jj < k ? jj : jj + 1;
            const c32 x = Z[z_plane(r, j, K, R, zblk) * TF + tf];
            const c32 ww = make_float2(wf[M + jj].x, sgn * wf[M + jj].y);
            ar = fmaf(ww.x, x.x, fmaf(-ww.y, x.y, ar));
            ai = fmaf(ww.x, x.y, fmaf(ww.y, x.x, ai));
        }
        out[g * TF + tf] = make_float2(ar, ai);
    }
}

// Shapes outside the (M, KR) template table (P > 8).  With P up to 16 the filter of a bin is as large as the data it
// multiplies, so re-reading it per (t, f) element (the flat mapping above) doubles the L1/L2 traffic: here a lane owns one
// BIN for a run of frames and keeps the bin's filter in registers.  M is a template parameter (the node's own M-vector is
// one unrolled, vectorised load); the remote count KR is a run-time value walked by a fully unrolled, wave-uniformly
// predicated loop (at most 15 remote rows).  One wave per workgroup: 64 consecutive bins x frames [t0, t1).
#ifndef DISCO_APPLY_XCD
#define DISCO_APPLY_XCD 8             // XCDs the workgroup ids of k_apply_m are dealt over
#endif
template <int M>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64) void k_apply_m(const c32* __restrict__ X, const c32* __restrict__ Z,
                                                 const c32* __restrict__ w, c32* __restrict__ out, int KR,
                                                 int K, int T, int F, int conj_w, int tiles, int t_chunks, int Kl, int k0, int zblk,
                                                 long long R) {
    const int P = M + KR;
    // Which workgroup does what: the Kl nodes of a room read the same K - 1 remote rows of a (tile, frame chunk), so they are made
    // neighbours on ONE XCD (one L2), as in k_cov_split_lds: the hardware deals consecutive workgroup ids round-robin to the 8
    // XCDs, hence id b is logical item (b % 8) * (grid / 8) + b / 8 (the grid is padded to a multiple of 8), and logical items run
    // node-fastest.  (With node-major ids node k's tile t sits on XCD (k + t) % 8: every remote row crossed the fabric K - 1
    // times -- PMC at C5: 32.1 GB read per launch against 18.5 GB of distinct bytes.)
    const long long n_items = R * Kl * (long long)tiles * t_chunks;
    long long item = (long long)(blockIdx.x % DISCO_APPLY_XCD) * (gridDim.x / DISCO_APPLY_XCD) + blockIdx.x / DISCO_APPLY_XCD;
    if (item >= n_items) return;
    const int kl = (int)(item % Kl);
    item /= Kl;
    const int tc = (int)(item % t_chunks);
    item /= t_chunks;
    const int tile = (int)(item % tiles);
    const long long g = (item / tiles) * Kl + kl;
    const int f = tile * 64 + (int)threadIdx.x;
    if (f >= F) return;
    const int t_len = (T + t_chunks - 1) / t_chunks;
    const int t0 = tc * t_len, t1 = min(T, t0 + t_len);
    const long long r = g / Kl;
    const int k = k0 + (int)(g % Kl);
    const long long TF = (long long)T * F;
    const float sgn = conj_w ? -1.f : 1.f;
    const c32* wf = w + (g * F + f) * (long long)P;
    c32 wl[M], wr[15];
#pragma unroll
    for (int i = 0; i < M; ++i) wl[i] = make_float2(wf[i].x, sgn * wf[i].y);
#pragma unroll
    for (int jj = 0; jj < 15; ++jj) wr[jj] = jj < KR ? make_float2(wf[M + jj].x, sgn * wf[M + jj].y) : make_float2(0.f, 0.f);

    for (int t = t0; t < t1; ++t) {
        const long long tf = (long long)t * F + f;
        const c32* xp = X + (g * TF + tf) * M;
        c32 x[M];
#pragma unroll
        for (int i = 0; i < M; ++i) x[i] = xp[i];
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            ar = fmaf(wl[i].x, x[i].x, fmaf(-wl[i].y, x[i].y, ar));
            ai = fmaf(wl[i].x, x[i].y, fmaf(wl[i].y, x[i].x, ai));
        }
#pragma unroll
        for (int jj = 0; jj < 15; ++jj) {
            if (jj < KR) {
                const int j = jj < k ? jj : jj + 1;
                const c32 z = Z[z_plane(r, j, K, R, zblk) * TF + tf];
                ar = fmaf(wr[jj].x, z.x, fmaf(-wr[jj].y, z.y, ar));
                ai = fmaf(wr[jj].x, z.y, fmaf(wr[jj].y, z.x, ai));
            }
        }
        out[g * TF + tf] = make_float2(ar, ai);
    }
}

// The same pass for M = 4 / 8 with the node's spectra fetched as CONTIGUOUS 16-byte granules (lane i of load r takes granule
// r * 64 + i of the tile's frame: 1 KiB per wave-load) instead of "every lane its own bin's M values" (16 bytes every 8 M bytes: one
// memory request per lane; a timing-only build with contiguous loads measured 4.57 instead of 5.5 ms per C5 launch).  The granules
// go through a wave-private LDS tile -- written in load order, read back bin-major, XOR-swizzled so that both directions are
// conflict-free (position (bin, p') holds granule p' ^ swz(bin), as in k_room_cov_dma) -- and the arithmetic is k_apply_m's,
// bit for bit.  (A first form that kept the granules where they landed and summed the MH lanes of a bin with DPP moves spread
// the remote rows over those lanes: 8 four-segment z loads per frame instead of 7 contiguous ones, 6.04 ms.)
// KRT: the number of remote rows the loop is unrolled for, >= KR (1 / 3 / 7 / 15); rows beyond KR re-read row KR - 1 with a zero tap, so that
// every load of a frame -- the granules and the remote rows -- is issued unconditionally, one frame ahead of its use, and the loop waits once.
template <int M, int KRT>
__global__ DISCO_KERNEL_ALIGN __launch_bounds__(64) void k_apply_mq(const c32* __restrict__ X, const c32* __restrict__ Z,
                                                  const c32* __restrict__ w, c32* __restrict__ out, int KR,
                                                  int K, int T, int F, int conj_w, int tiles, int t_chunks, int Kl, int k0, int zblk,
                                                  long long R) {
    static_assert(M == 4 || M == 8, "whole 256-byte bank rows per group of bins");
    constexpr int MH = M / 2, BPR = 16 / MH;            // granules per bin; bins per 256-byte bank row
    __shared__ float4 sx[64 * MH];
    const int P = M + KR;
    const long long n_items = R * Kl * (long long)tiles * t_chunks;
    long long item = (long long)(blockIdx.x % DISCO_APPLY_XCD) * (gridDim.x / DISCO_APPLY_XCD) + blockIdx.x / DISCO_APPLY_XCD;
    if (item >= n_items) return;
    const int kl = (int)(item % Kl);
    item /= Kl;
    const int tc = (int)(item % t_chunks);
    item /= t_chunks;
    const int tile = (int)(item % tiles);
    const long long g = (item / tiles) * Kl + kl;
    const int lane = (int)threadIdx.x;
    const int f = tile * 64 + lane;
    const bool live = f < F;
    const int fc = live ? f : F - 1;                    // the ragged last tile: spare lanes work on bin F - 1 and store nothing
    const int t_len = (T + t_chunks - 1) / t_chunks;
    const int t0 = tc * t_len, t1 = min(T, t0 + t_len);
    if (t0 >= t1) return;
    const long long r_ = g / Kl;
    const int k = k0 + (int)(g % Kl);
    const long long TF = (long long)T * F;
    const float sgn = conj_w ? -1.f : 1.f;
    const c32* wf = w + (g * F + fc) * (long long)P;
    c32 wl[M], wr[KRT];
#pragma unroll
    for (int i = 0; i < M; ++i) wl[i] = make_float2(wf[i].x, sgn * wf[i].y);
    const c32* zrow[KRT];                               // (wave-uniform)
#pragma unroll
    for (int jj = 0; jj < KRT; ++jj) {
        wr[jj] = jj < KR ? make_float2(wf[M + jj].x, sgn * wf[M + jj].y) : make_float2(0.f, 0.f);
        const int jc = jj < KR ? jj : KR - 1;
        zrow[jj] = Z + z_plane(r_, jc < k ? jc : jc + 1, K, R, zblk) * TF + fc;
    }
    // loader: granule r * 64 + lane of the tile's frame = (bin b, granule lane % MH), clamped like fc; LDS position of that granule
    int lgo[MH], lpos[MH];
#pragma unroll
    for (int r = 0; r < MH; ++r) {
        const int gi = r * 64 + lane, b = gi / MH, pp = gi % MH;
        lgo[r] = min(tile * 64 + b, F - 1) * MH + pp;
        lpos[r] = b * MH + (pp ^ ((b / BPR) % MH));
    }
    const int swz = (lane / BPR) % MH;
    const float4* Xg = reinterpret_cast<const float4*>(X + g * TF * M);
    float q[MH][4];                                     // (scalars on purpose: a float4 copied global -> private -> LDS stays a memcpy through scratch)
    c32 zq[KRT];
    auto fetch = [&](int t) {
        const long long tF = (long long)t * F;
#pragma unroll
        for (int r = 0; r < MH; ++r) {
            const float4 v = Xg[tF * MH + lgo[r]];
            q[r][0] = v.x;
            q[r][1] = v.y;
            q[r][2] = v.z;
            q[r][3] = v.w;
        }
#pragma unroll
        for (int jj = 0; jj < KRT; ++jj) zq[jj] = zrow[jj][tF];
    };
    fetch(t0);
    for (int t = t0; t < t1; ++t) {
#pragma unroll
        for (int r = 0; r < MH; ++r) sx[lpos[r]] = make_float4(q[r][0], q[r][1], q[r][2], q[r][3]);
        c32 z[KRT];
#pragma unroll
        for (int jj = 0; jj < KRT; ++jj) z[jj] = zq[jj];
        DISCO_LDS_RAW();
        fetch(min(t + 1, t1 - 1));                      // the next frame is on its way while this one is filtered (the last one again at the end)
        c32 x[M];
#pragma unroll
        for (int pp = 0; pp < MH; ++pp) {
            const float4 v = sx[lane * MH + (pp ^ swz)];
            x[2 * pp] = make_float2(v.x, v.y);
            x[2 * pp + 1] = make_float2(v.z, v.w);
        }
        DISCO_LDS_RAW();                                // the reads have returned before the next frame overwrites the tile
        float ar = 0.f, ai = 0.f;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            ar = fmaf(wl[i].x, x[i].x, fmaf(-wl[i].y, x[i].y, ar));
            ai = fmaf(wl[i].x, x[i].y, fmaf(wl[i].y, x[i].x, ai));
        }
#pragma unroll
        for (int jj = 0; jj < KRT; ++jj) {
            ar = fmaf(wr[jj].x, z[jj].x, fmaf(-wr[jj].y, z[jj].y, ar));
            ai = fmaf(wr[jj].x, z[jj].y, fmaf(wr[jj].y, z[jj].x, ai));
        }
        if (live) out[g * TF + (long long)t * F + fc] = make_float2(ar, ai);
    }
}

// w_loc[g][f][0:M] <- w_glo[g][f][0:M]  (the local part of a P-entry filter; iterated scheme)
static __global__ void k_filter_head(const c32* __restrict__ w_glo, c32* __restrict__ w_loc, long long n_bins, int M, int P) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_bins * M; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / M;
        w_loc[i] = w_glo[b * P + (int)(i % M)];
    }
}

// zn = Y[ref] - z  (tango.py:376)
static __global__ void k_noise_residual(const c32* __restrict__ X, const c32* __restrict__ z, c32* __restrict__ zn,
                                 long long n, int M, int ref) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const c32 x = X[i * M + ref];
        zn[i] = make_float2(x.x - z[i].x, x.y - z[i].y);
    }
}

// ---- small elementwise helpers of the reference-output path (disco_tango_reference) ---------------------------------
// the sender-side variants of mask_for_z (tango.py:396-405): zs = m z, zn = (1 - m) z
static __global__ void k_mask_rows(const c32* __restrict__ z, const float* __restrict__ m, c32* __restrict__ zs, c32* __restrict__ zn,
                            long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const c32 v = z[i];
        const float a = m[i], b = 1.f - m[i];
        zs[i] = make_float2(a * v.x, a * v.y);
        zn[i] = make_float2(b * v.x, b * v.y);
    }
}
// plane[i] = X[i*M + ch]   ('use_oracle_refs': the remote rows are the oracle images at the reference microphone, tango.py:406-407)
static __global__ void k_pick_channel(const c32* __restrict__ X, c32* __restrict__ plane, long long n, int M, int ch) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        plane[i] = X[i * M + ch];
}
static __global__ void k_fill_f32(float* __restrict__ p, float v, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

// GRU cell, pointwise part (gate order r, z, n as torch.nn.GRU): the two matrix products are GEMMs of the caller,
//   r = sigm(gi_r + gh_r), z = sigm(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h
// gi: rows of 3H floats at stride gi_stride (a time slice of the all-steps input projection); gh [n][3H] or NULL with gh_bias
// [3H] (first step: h = 0, the recurrent product is its bias); h_prev [n][H] or NULL (= 0); h_out [n][H].
static __global__ void k_gru_gates(const float* __restrict__ gi, long long gi_stride, const float* __restrict__ gh, const float* __restrict__ gh_bias,
                            const float* __restrict__ h_prev, float* __restrict__ h_out, long long n, int H) {
    const long long total = n * H;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / H;
        const int c = (int)(i - row * H);
        const float* g = gi + row * gi_stride;
        float hr, hz, hn;
        if (gh) {
            const float* q = gh + row * 3 * H;
            hr = q[c];
            hz = q[H + c];
            hn = q[2 * H + c];
        } else {
            hr = gh_bias[c];
            hz = gh_bias[H + c];
            hn = gh_bias[2 * H + c];
        }
        const float r = 1.f / (1.f + expf(-(g[c] + hr)));
        const float z = 1.f / (1.f + expf(-(g[H + c] + hz)));
        const float nn = tanhf(g[2 * H + c] + r * hn);
        const float hp = h_prev ? h_prev[i] : 0.f;
        h_out[i] = (1.f - z) * nn + z * hp;
    }
}

// MaxPool2d((1, 4)) over the last axis (floor mode) of x [B][C][rows_per_ch][row_len], plus the convolution's per-channel bias
// (a constant commutes with the maximum, and adding it here saves a read-modify-write pass over the 4x larger input):
// out[r][q] = max x[r][4q .. 4q+3] + bias[channel of row r],  q < row_len / 4
static __global__ void k_maxpool_last4(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ out, long long n_rows,
                                int row_len, int rows_per_ch, int C) {
    const int nq = row_len / 4;
    const long long total = n_rows * nq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / nq;
        const int q = (int)(i - r * nq);
        const float* p = x + r * row_len + 4 * q;
        const float b = bias ? bias[(r / rows_per_ch) % C] : 0.f;
        // torch.nn.MaxPool2d propagates NaN, fmaxf drops it: the sum is NaN iff one of the four is
        const float mx = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
        const float any = (p[0] + p[1]) + (p[2] + p[3]);
        out[i] = (any != any ? any : mx) + b;
    }
}

// Streaming kernel of KNOWN traffic with the access width of the STFT kernels' sample reads: every lane moves ONE dword per
// instruction (256 B per wave load), UNROLL independent loads in flight.  write = 0: read-only (one float per workgroup is stored).
// Calibrates the HBM counters (FETCH_SIZE / WRITE_SIZE) for 4-byte-per-lane loads, which torch's 16-byte-vectorised elementwise
// kernels do not exercise (tools/pmc_traffic.py).
template <bool write>
__global__ __launch_bounds__(256) void k_selftest_stream(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    constexpr int UNROLL = 8;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float acc = 0.f;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (write) dst[i + u * stride] = v[u];
            else acc += v[u];
        }
    }
    for (; i < n; i += stride) {
        if (write) dst[i] = src[i];
        else acc += src[i];
    }
    if (!write && acc == 123.456f) dst[blockIdx.x] = acc;      // keeps the loads alive; practically never taken
}

// The recurrent layer's input windows (crnn.py:59 `.view`): out[(b T + t) * n_keep + e] = feat[b][c][t + w][fy] with
// e = (c W + w) 4 + fy < n_keep -- the leading n_keep floats of window t's (C, W, 4) block, moved 16 bytes at a time.
// feat [B][C][Tp][4] (Tp >= T + W - 1), n_keep a multiple of 4.
static __global__ void k_crnn_windows(const float4* __restrict__ feat, float4* __restrict__ out, long long B, int C, int Tp, int T, int W,
                               int n_keep4) {
    const long long total = B * T * n_keep4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e4 = (int)(i % n_keep4);
        const long long bt = i / n_keep4;
        const int t = (int)(bt % T);
        const long long b = bt / T;
        const int c = e4 / W, w = e4 - c * W;
        out[i] = feat[(b * C + c) * Tp + t + w];
    }
}

}  // namespace disco
