// Image-source room impulse responses for shoebox rooms -- the generator in front of disco_rir_convolve.
// The reference obtains its RIRs from pyroomacoustics (dataset_generation/gen_disco/convolve_signals.py:243-246:
// pra.ShoeBox(dims, fs, max_order=20, absorption=alpha); :94-95 image_source_model + compute_rir).  pyroomacoustics is a
// third-party C++/Python package, absent here and unpinned in the reference: this is the published algorithm (Allen &
// Berkley 1979) with that package's conventions as documented --
//   * image (nx, ny, nz), |nx| + |ny| + |nz| <= max_order: coordinate n L + s (n even) or n L + (L - s) (n odd);
//   * every reflection multiplies the amplitude by sqrt(1 - absorption); amplitude / (4 pi d) at distance d;
//   * fractional delays by an 81-tap Hann-windowed sinc centred on the delay, the whole response shifted by 40 samples.
// Parity: against oracle/ism_oracle.py (the same formulas in float64); "parity unpinned" with respect to pyroomacoustics.
// One workgroup per (room, source, microphone): the response is accumulated in LDS with float atomics.
#pragma once
#include "common.h"

namespace disco {

constexpr int ISM_THREADS = 256;
constexpr int ISM_MAX_LEN = 8192;           // taps held in LDS
constexpr int ISM_FDL = 81, ISM_FDL2 = 40;

static __global__ __launch_bounds__(ISM_THREADS) void k_ism_rir(const float* __restrict__ dims, const float* __restrict__ absorption,
                                                          const float* __restrict__ src, const float* __restrict__ mic,
                                                          int S, int Q, int max_order, float fs, float c_sound,
                                                          float* __restrict__ rir, int Lh) {
    __shared__ float h[ISM_MAX_LEN];
    __shared__ float win[ISM_FDL];                        // np.hanning(81)
    const long long id = blockIdx.x;
    const int q = (int)(id % Q), s = (int)((id / Q) % S);
    const long long room = id / ((long long)S * Q);
    for (int i = threadIdx.x; i < Lh; i += ISM_THREADS) h[i] = 0.f;
    if (threadIdx.x < ISM_FDL) win[threadIdx.x] = (float)(0.5 - 0.5 * cos(2.0 * 3.14159265358979323846 * (double)threadIdx.x / (double)(ISM_FDL - 1)));
    __syncthreads();
    const double Lx = dims[room * 3 + 0], Ly = dims[room * 3 + 1], Lz = dims[room * 3 + 2];
    const float* sp = src + (room * S + s) * 3;
    const float* mp = mic + (room * Q + q) * 3;
    const double sx = sp[0], sy = sp[1], sz = sp[2], mx = mp[0], my = mp[1], mz = mp[2];
    const double refl = sqrt(1.0 - (double)absorption[room]);
    const int W = 2 * max_order + 1;
    const long long n_img = (long long)W * W * W;
    const double PI = 3.14159265358979323846;
    for (long long ii = threadIdx.x; ii < n_img; ii += ISM_THREADS) {
        const int nx = (int)(ii % W) - max_order, ny = (int)((ii / W) % W) - max_order, nz = (int)(ii / ((long long)W * W)) - max_order;
        const int order = abs(nx) + abs(ny) + abs(nz);
        if (order > max_order) continue;
        const double ix = nx * Lx + ((nx & 1) ? Lx - sx : sx);
        const double iy = ny * Ly + ((ny & 1) ? Ly - sy : sy);
        const double iz = nz * Lz + ((nz & 1) ? Lz - sz : sz);
        const double d = sqrt((ix - mx) * (ix - mx) + (iy - my) * (iy - my) + (iz - mz) * (iz - mz));
        const double tau = d / (double)c_sound * (double)fs;
        const int ip = (int)floor(tau);
        const double fp = tau - (double)ip;
        if (ip >= Lh) continue;                                  // entirely beyond the response (the shift only adds)
        // per-image quantities in float64 (a 1e-3-sample error of the fractional delay would already cost 1e-3 in the taps),
        // the 81 taps in float32
        const float amp = (float)(pow(refl, (double)order) / (4.0 * PI * d));
        const float sfp = (float)sin(PI * fp);
        const float inv_pi = 0.318309886183790671538f;
        for (int kk = -ISM_FDL2; kk <= ISM_FDL2; ++kk) {
            const int idx = ip + kk + ISM_FDL2;                 // response shifted by ISM_FDL2 samples
            if (idx < 0 || idx >= Lh) continue;
            // k - fp in float64 BEFORE rounding: next to the delay (|k - fp| << 1) a float32 subtraction would cancel
            const float x = (float)((double)kk - fp);
            // sinc(k - fp) = sin(pi (k - fp)) / (pi (k - fp)) = -(-1)^k sin(pi fp) / (pi (k - fp))
            const float sinc = (x == 0.f) ? 1.f : ((kk & 1) ? sfp : -sfp) * inv_pi / x;
            atomicAdd(&h[idx], amp * win[kk + ISM_FDL2] * sinc);
        }
    }
    __syncthreads();
    float* o = rir + id * (long long)Lh;
    for (int i = threadIdx.x; i < Lh; i += ISM_THREADS) o[i] = h[i];
}

}  // namespace disco
