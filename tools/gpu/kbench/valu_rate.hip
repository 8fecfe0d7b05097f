// VALU issue-rate probe for gfx950: cycles per wave64 instruction of scalar and packed f32 forms, with and without
// op_sel / neg modifiers, at 1 / 2 / 4 waves per SIMD.  Standalone: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int V>
__global__ __launch_bounds__(256) void k_rate(float* out, long long* cyc, int iters) {
    v2f a[8], b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, 1.f + i};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (V == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x));
                REP8(X)
#undef X
            } else if constexpr (V == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (V == 2) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(a[i]) : "v"(b), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (V == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if constexpr (V == 4) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(a[i]) : "v"(c));
                REP8(X)
#undef X
            } else if constexpr (V == 5) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[0,0] op_sel_hi:[0,1]" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (V == 6) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
                REP8(X)
#undef X
            } else if constexpr (V == 7) {      // scalar and packed alternating
#define X(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %4, %5" : "+v"(a[i].x), "+v"(a[(i + 4) & 7]) : "v"(b.x), "v"(c.x), "v"(b), "v"(c));
                X(0) X(1) X(2) X(3)
#undef X
            } else if constexpr (V == 8) {      // packed fma with one operand pair in SGPRs
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(v2f{1.0001f, 0.9999f}), "v"(c));
                REP8(X)
#undef X
            } else if constexpr (V == 9) {      // v_pk_mul_f32 plain
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                REP8(X)
#undef X
            } else if constexpr (V == 10) {     // v_mul_f32
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x));
                REP8(X)
#undef X
            } else if constexpr (V == 11) {     // dependent chain of packed adds (latency)
                asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n"
                             "v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(c));
            } else if constexpr (V == 12) {     // dependent chain of scalar adds (latency)
                asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                             "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a[0].x) : "v"(c.x));
            }
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[(long long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* name, int instr_per_u) {
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;            // 256 CUs x wps workgroups of 4 waves (one per SIMD)
        float* out;
        long long* cyc;
        hipMalloc(&out, (size_t)blocks * 256 * 4);
        hipMalloc(&cyc, (size_t)blocks * 8);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k_rate<V>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rate<V>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        long long* h = (long long*)malloc((size_t)blocks * 8);
        hipMemcpy(h, cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (int i = 0; i < blocks; ++i) avg += (double)h[i];
        avg /= blocks;
        const double n_instr = (double)iters * 4 * instr_per_u;       // per wave
        printf("%-34s waves/SIMD %d : %8.3f ms  clock64/instr/wave %6.2f  -> %5.2f per SIMD-instr; wall ns/instr/SIMD %6.3f\n", name, wps, ms,
               avg / n_instr, avg / n_instr / wps, ms * 1e6 / (n_instr * wps));
        free(h);
        hipFree(out);
        hipFree(cyc);
    }
}

int main() {
    run<0>("v_fma_f32", 8);
    run<6>("v_add_f32", 8);
    run<10>("v_mul_f32", 8);
    run<1>("v_pk_fma_f32", 8);
    run<2>("v_pk_fma_f32 op_sel+neg", 8);
    run<8>("v_pk_fma_f32 sgpr pair", 8);
    run<3>("v_pk_add_f32", 8);
    run<4>("v_pk_add_f32 op_sel+neg", 8);
    run<9>("v_pk_mul_f32", 8);
    run<5>("v_pk_mul_f32 op_sel", 8);
    run<7>("v_fma_f32 + v_pk_fma_f32 alternating", 8);
    run<11>("v_pk_add_f32 dependent chain", 8);
    run<12>("v_add_f32 dependent chain", 8);
    return 0;
}
