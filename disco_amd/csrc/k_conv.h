// The step before the path: reverberating dry signals with room impulse responses,
//     out[i][c] = np.convolve(dry[i], rir[i][c])[:Lout]          dataset_generation/gen_disco/convolve_signals.py:160-163
// (pyroomacoustics' room.simulate, :94-97, is the same operation on the target), batched over (signal, channel).
//
// Uniformly partitioned overlap-save with the wave FFT of fft.h (N = 1024, 512 new samples per block):
//   k_conv_spectra<true>   X[i][j]    = FFT(dry[i][(j-1) 512 .. (j+1) 512))              one wave per block
//   k_conv_spectra<false>  H[i][c][p] = FFT([rir[i][c][p 512 .. (p+1) 512) ; 0 ... 0])   one wave per partition
//   k_conv_mac_ifft        Y[b] = sum_p X[i][b-p] H[i][c][p]  ->  inverse FFT  ->  its last 512 samples are out[b 512 ...)
// Only the 513 non-redundant bins of the real signals' spectra are stored.  In the third kernel a workgroup owns one
// (signal, channel): the channel's partition spectra sit in LDS for the whole signal, each wave walks groups of CV_NB
// consecutive output blocks with their accumulators in registers, so that an input spectrum X[j] is loaded once per
// group and used for up to CV_NB (block, partition) products.
#pragma once
#include "fft.h"

namespace disco {

constexpr int CV_N = 1024, CV_B = 512, CV_F = CV_N / 2 + 1;
constexpr int CV_E = FftPlan<CV_N>::E;        // 16 points per lane
constexpr int CV_EH = CV_E / 2;               // 8 stored bins per lane (+ bin 512)
constexpr int CV_WAVES = 4;
constexpr int CV_NB = 4;                      // output blocks per wave group

// rows of `len` samples -> spectra S[row][blk][CV_F]
template <bool OVERLAP>
__global__ __launch_bounds__(64 * CV_WAVES) void k_conv_spectra(const float* __restrict__ x, long long len, int n_blocks,
                                                                 long long n_items, c32* __restrict__ S,
                                                                 const c32* __restrict__ tw) {
    __shared__ c32 buf[CV_WAVES][fft_buf_len<CV_N>()];
    const int lane = threadIdx.x & 63, w = wave_id();
    WaveTw<CV_N> wtw;
    wtw.init(tw, lane);
    for (long long it = (long long)blockIdx.x * CV_WAVES + w; it < n_items; it += (long long)gridDim.x * CV_WAVES) {
        const long long row = it / n_blocks;
        const int b = (int)(it % n_blocks);
        const float* xr = x + row * len;
        const long long base = OVERLAP ? (long long)(b - 1) * CV_B : (long long)b * CV_B;
        c32 v[CV_E];
#pragma unroll
        for (int e = 0; e < CV_E; ++e) {
            const int n = lane + 64 * e;
            const long long idx = base + n;
            const bool in = idx >= 0 && idx < len && (OVERLAP || n < CV_B);
            const long long ic = idx < 0 ? 0 : (idx >= len ? len - 1 : idx);
            const float val = xr[ic];
            v[e] = make_float2(in ? val : 0.f, 0.f);
        }
        fft_wave<CV_N>(v, wtw, buf[w], lane);
        c32* So = S + it * CV_F;
#pragma unroll
        for (int e = 0; e < CV_EH; ++e) So[lane + 64 * e] = v[e];
        if (lane == 0) So[CV_N / 2] = v[CV_EH];
    }
}

template <int PMAX>
struct alignas(16) ConvShared {
    c32 H[PMAX][CV_F + 3];                    // partition spectra of this (signal, channel)
    c32 buf[CV_WAVES][fft_buf_len<CV_N>()];
};

template <int PMAX>
__global__ __launch_bounds__(64 * CV_WAVES) void k_conv_mac_ifft(const c32* __restrict__ X, const c32* __restrict__ Hs,
                                                                  float* __restrict__ out, const c32* __restrict__ tw,
                                                                  int n_ch, int n_blocks, int P, int Lout) {
    __shared__ ConvShared<PMAX> sh;
    const int lane = threadIdx.x & 63, w = wave_id();
    const long long ic = blockIdx.x;                       // (signal, channel)
    const long long i = ic / n_ch;
    {
        const c32* src = Hs + ic * (long long)P * CV_F;
        for (int q = threadIdx.x; q < P * CV_F; q += 64 * CV_WAVES) sh.H[q / CV_F][q % CV_F] = src[q];
    }
    WaveTw<CV_N> wtw;
    wtw.init(tw, lane);
    __syncthreads();
    const c32* Xi = X + i * (long long)n_blocks * CV_F;
    float* og = out + ic * (long long)Lout;
    c32* buf = sh.buf[w];
    const int n_groups = (n_blocks + CV_NB - 1) / CV_NB;
    for (int g = w; g < n_groups; g += CV_WAVES) {
        const int b0 = g * CV_NB;
        c32 acc[CV_NB][CV_EH + 1];
#pragma unroll
        for (int bb = 0; bb < CV_NB; ++bb)
#pragma unroll
            for (int e = 0; e <= CV_EH; ++e) acc[bb][e] = make_float2(0.f, 0.f);
        const int j_lo = max(0, b0 - P + 1), j_hi = min(n_blocks - 1, b0 + CV_NB - 1);
        for (int j = j_lo; j <= j_hi; ++j) {
            const c32* Xp = Xi + (long long)j * CV_F;
            c32 x[CV_EH + 1];
#pragma unroll
            for (int e = 0; e < CV_EH; ++e) x[e] = Xp[lane + 64 * e];
            x[CV_EH] = Xp[CV_N / 2];
#pragma unroll
            for (int bb = 0; bb < CV_NB; ++bb) {
                const int p = b0 + bb - j;                 // wave-uniform
                if (p >= 0 && p < P) {
#pragma unroll
                    for (int e = 0; e < CV_EH; ++e) {
                        const c32 h = sh.H[p][lane + 64 * e];
                        acc[bb][e].x = fmaf(x[e].x, h.x, fmaf(-x[e].y, h.y, acc[bb][e].x));
                        acc[bb][e].y = fmaf(x[e].x, h.y, fmaf(x[e].y, h.x, acc[bb][e].y));
                    }
                    const c32 h = sh.H[p][CV_N / 2];
                    acc[bb][CV_EH].x = fmaf(x[CV_EH].x, h.x, fmaf(-x[CV_EH].y, h.y, acc[bb][CV_EH].x));
                    acc[bb][CV_EH].y = fmaf(x[CV_EH].x, h.y, fmaf(x[CV_EH].y, h.x, acc[bb][CV_EH].y));
                }
            }
        }
        // inverse real FFT of every block: y = Re(FFT(conj Z)) / N with Z the Hermitian extension of acc
#pragma unroll
        for (int bb = 0; bb < CV_NB; ++bb) {
            const int b = b0 + bb;
            if (b < n_blocks) {                            // wave-uniform
                DISCO_LDS_WAR();
#pragma unroll
                for (int e = 0; e < CV_EH; ++e) {
                    const int f = lane + 64 * e;
                    const c32 a = acc[bb][e];
                    buf[fft_pad<CV_N>(f)] = make_float2(a.x, -a.y);                       // conj(Z[f])
                    if (f != 0) buf[fft_pad<CV_N>(CV_N - f)] = a;                          // conj(Z[N-f]) = Z[f]
                }
                if (lane == 0) buf[fft_pad<CV_N>(CV_N / 2)] = make_float2(acc[bb][CV_EH].x, -acc[bb][CV_EH].y);
                DISCO_LDS_RAW();
                c32 v[CV_E];
#pragma unroll
                for (int e = 0; e < CV_E; ++e) v[e] = buf[fft_pad<CV_N>(lane + 64 * e)];
                fft_wave<CV_N>(v, wtw, buf, lane);
#pragma unroll
                for (int e = CV_EH; e < CV_E; ++e) {                                       // samples 512 .. 1023 of the window
                    const long long pos = (long long)b * CV_B + lane + 64 * (e - CV_EH);
                    if (pos < Lout) og[pos] = v[e].x * (1.0f / CV_N);
                }
            }
        }
    }
}

}  // namespace disco
