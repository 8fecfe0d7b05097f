// Shared device helpers for the DISCO MWF kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "../../include/disco_hip.h"

namespace disco {

typedef float2 c32;    // interleaved complex64, bit-compatible with disco_c32 / numpy complex64
typedef double2 c64;   // complex128

constexpr int WAVE = 64;

__device__ __forceinline__ c32 cmul(c32 a, c32 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c32 cconj(c32 a) { return make_float2(a.x, -a.y); }
// multiply by -i / +i
__device__ __forceinline__ c32 cmul_mi(c32 a) { return make_float2(a.y, -a.x); }
__device__ __forceinline__ c32 cmul_pi(c32 a) { return make_float2(-a.y, a.x); }

__device__ __forceinline__ c64 zmul(c64 a, c64 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
__device__ __forceinline__ c64 zmulc(c64 a, c64 b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
__device__ __forceinline__ c64 zadd(c64 a, c64 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c64 zsub(c64 a, c64 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c64 zscale(c64 a, double s) { return make_double2(a.x * s, a.y * s); }

// reflect / zero padded sample fetch: p is the index into the un-padded signal of length L
__device__ __forceinline__ float load_padded(const float* __restrict__ x, int p, int L, int pad_mode) {
    if (p < 0) {
        if (pad_mode != DISCO_PAD_REFLECT) return 0.f;
        p = -p;
    } else if (p >= L) {
        if (pad_mode != DISCO_PAD_REFLECT) return 0.f;
        p = 2 * (L - 1) - p;
    }
    // np.pad(reflect) needs L > n_fft/2; clamp keeps short inputs in bounds instead of faulting
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    return x[p];
}

}  // namespace disco
