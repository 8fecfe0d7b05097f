// hipemu -- a tiny CPU *emulator* of the HIP execution model, for TESTS ONLY.
//
// Purpose: run the kernel sources of disco_amd/csrc/ unmodified on a machine without a GPU, so that
// `pytest -m "not gpu"` can check kernel LOGIC (indexing, FFT passes, reductions, the Jacobi solver)
// against the oracle at toy sizes.  It is NOT a product path and NOT a fallback: the shipped library
// (disco_amd/lib/libdisco_hip.so) is built by hipcc for gfx950 only and the Python package refuses to run
// without it.  Only tests/emu_build.py compiles against this header, into tests/_emu/.
//
// Model: every GPU thread of a block is a user-level fiber (ucontext) of one OS thread, switched only at barriers and
// shuffles; __syncthreads() / wave barriers are generation-counting barriers over the live fibers; a wave is 64
// consecutive threads; shuffles go through a per-wave slot array; blocks are handed out to one OS thread per host core
// and `__shared__` becomes `static thread_local` (one LDS image per OS thread).  gcc, -pthread.
#pragma once
#include <ucontext.h>

#include <atomic>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIPEMU 1
#define DISCO_CONSUME(x) ((void)(x))      // register-level scheduling pin of the GPU build: nothing to emulate

struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

namespace hipemu {
// A barrier of fibers: trips when every LIVE participant has arrived (threads that returned no longer count, as on the GPU).
struct Bar {
    int alive = 0, count = 0;
    unsigned gen = 0;
};
struct WaveState {
    Bar bar;
    uint64_t slot[64];
    int n;                       // lanes of this wave (valid shuffle sources)
    unsigned char xbuf[64][512]; // per-lane exchange area of gather_lane / gather_row16 (DPP broadcast forms, csrc/dpp64.h)
};
struct BlockState {
    Bar bar;
    std::vector<WaveState> waves;
};
struct Fiber {
    ucontext_t ctx;
    char* stack = nullptr;
    bool done = false;
    Bar* wait_bar = nullptr;     // barrier the fiber sleeps on (runnable when its generation moves on)
    unsigned wait_gen = 0;
    uint3 tidx;
};
struct Worker {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    BlockState bs;
    unsigned cur = 0;
    const std::function<void()>* fn = nullptr;
    ~Worker() {
        for (auto& f : fibers) std::free(f.stack);
    }
};
constexpr size_t kStackBytes = 256 * 1024;
inline thread_local uint3 t_threadIdx, t_blockIdx;
inline thread_local dim3 t_blockDim, t_gridDim;
inline thread_local BlockState* t_block = nullptr;
inline thread_local WaveState* t_wave = nullptr;
inline thread_local int t_lane = 0;
inline thread_local Worker* t_worker = nullptr;

inline void fiber_main() {
    Worker* w = t_worker;
    Fiber& f = w->fibers[w->cur];
    (*w->fn)();
    f.done = true;               // uc_link returns to the scheduler
}

inline void bar_wait(Bar& b) {
    Worker* w = t_worker;
    const unsigned g = b.gen;
    if (++b.count >= b.alive) {
        b.count = 0;
        ++b.gen;
        return;
    }
    Fiber& f = w->fibers[w->cur];
    f.wait_bar = &b;
    f.wait_gen = g;
    swapcontext(&f.ctx, &w->sched);
}

// Runs one block: every GPU thread is a fiber of THIS OS thread, switched at barriers / shuffles only.
inline void run_block(Worker& w, dim3 block, unsigned nthreads) {
    BlockState& bs = w.bs;
    const unsigned nwaves = (nthreads + 63) / 64;
    bs.bar = Bar{};
    bs.bar.alive = (int)nthreads;
    bs.waves.resize(nwaves);
    for (unsigned v = 0; v < nwaves; ++v) {
        bs.waves[v].n = (int)std::min(64u, nthreads - 64 * v);
        bs.waves[v].bar = Bar{};
        bs.waves[v].bar.alive = bs.waves[v].n;
    }
    if (w.fibers.size() < nthreads) w.fibers.resize(nthreads);
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = w.fibers[t];
        if (!f.stack) f.stack = (char*)std::malloc(kStackBytes);
        f.done = false;
        f.wait_bar = nullptr;
        f.tidx = uint3{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = &w.sched;
        makecontext(&f.ctx, (void (*)())fiber_main, 0);
    }
    unsigned remaining = nthreads;
    while (remaining) {
        bool progressed = false;
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = w.fibers[t];
            if (f.done) continue;
            if (f.wait_bar) {
                if (f.wait_bar->gen == f.wait_gen) continue;      // still asleep
                f.wait_bar = nullptr;
            }
            progressed = true;
            w.cur = t;
            t_threadIdx = f.tidx;
            t_wave = &bs.waves[t / 64];
            t_lane = (int)(t % 64);
            swapcontext(&w.sched, &f.ctx);
            if (f.done) {
                // a thread that returned no longer takes part in barriers: release the ones it would have completed
                --remaining;
                Bar* bars[2] = {&bs.bar, &bs.waves[t / 64].bar};
                for (Bar* b : bars) {
                    --b->alive;
                    if (b->alive > 0 && b->count >= b->alive) {
                        b->count = 0;
                        ++b->gen;
                    }
                }
            }
        }
        if (!progressed) {
            std::fprintf(stderr, "hipemu: deadlock -- %u thread(s) of a block wait on a barrier the others never reach\n", remaining);
            std::abort();
        }
    }
}

// Blocks are independent: they are handed out to one OS thread per host core (`__shared__` is thread_local static storage,
// so every OS thread has its own LDS image, reused block after block).
inline void launch(dim3 grid, dim3 block, const std::function<void()>& fn) {
    const unsigned nthreads = block.x * block.y * block.z;
    const unsigned long long nblocks = (unsigned long long)grid.x * grid.y * grid.z;
    if (!nthreads || !nblocks) return;
    std::atomic<unsigned long long> next{0};
    auto work = [&]() {
        thread_local Worker worker;
        Worker& w = worker;
        t_worker = &w;
        t_block = &w.bs;
        t_blockDim = block;
        t_gridDim = grid;
        w.fn = &fn;
        for (;;) {
            const unsigned long long b = next.fetch_add(1);
            if (b >= nblocks) break;
            t_blockIdx = uint3{(unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((unsigned long long)grid.x * grid.y))};
            run_block(w, block, nthreads);
        }
    };
    unsigned hw = std::thread::hardware_concurrency();
    const unsigned nworkers = (unsigned)std::min<unsigned long long>(hw ? hw : 1, nblocks);
    if (nworkers <= 1) {
        work();
        return;
    }
    std::vector<std::thread> th;
    th.reserve(nworkers);
    for (unsigned t = 0; t < nworkers; ++t) th.emplace_back(work);
    for (auto& t : th) t.join();
}

template <class T>
inline T shfl_abs(T v, int src_abs) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t bits = 0;
    std::memcpy(&bits, &v, sizeof(T));
    t_wave->slot[t_lane] = bits;
    bar_wait(t_wave->bar);
    uint64_t r = t_wave->slot[(src_abs >= 0 && src_abs < t_wave->n) ? src_abs : t_lane];
    bar_wait(t_wave->bar);
    T out;
    std::memcpy(&out, &r, sizeof(T));
    return out;
}
// `bytes` (<= 512) of lane src_abs's `mine` -> out, for every lane of the wave at once (one rendezvous pair)
inline void gather_lane(const void* mine, size_t bytes, int src_abs, void* out) {
    std::memcpy(t_wave->xbuf[t_lane], mine, bytes);
    bar_wait(t_wave->bar);
    std::memcpy(out, t_wave->xbuf[(src_abs >= 0 && src_abs < t_wave->n) ? src_abs : t_lane], bytes);
    bar_wait(t_wave->bar);
}
// every lane's `mine` (bytes <= 32) of this lane's row of 16 -> out[16][bytes]
inline void gather_row16(const void* mine, size_t bytes, void* out) {
    std::memcpy(t_wave->xbuf[t_lane], mine, bytes);
    bar_wait(t_wave->bar);
    const int base = t_lane & ~15;
    for (int l = 0; l < 16; ++l)
        std::memcpy((unsigned char*)out + l * bytes, t_wave->xbuf[(base + l) < t_wave->n ? base + l : t_lane], bytes);
    bar_wait(t_wave->bar);
}
}  // namespace hipemu

#define threadIdx (hipemu::t_threadIdx)
#define blockIdx (hipemu::t_blockIdx)
#define blockDim (hipemu::t_blockDim)
#define gridDim (hipemu::t_gridDim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::bar_wait(hipemu::t_block->bar); }

template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    int base = hipemu::t_lane & ~(width - 1);
    return hipemu::shfl_abs(v, base + (src & (width - 1)));
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    int base = hipemu::t_lane & ~(width - 1);
    return hipemu::shfl_abs(v, base + ((hipemu::t_lane ^ mask) & (width - 1)));
}
template <class T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int base = hipemu::t_lane & ~(width - 1);
    int idx = (hipemu::t_lane & (width - 1)) + (int)delta;
    return hipemu::shfl_abs(v, idx < width ? base + idx : hipemu::t_lane);
}
static inline unsigned long long __ballot(int pred) {
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l) {
        int p = hipemu::shfl_abs(pred, l);
        if (l < hipemu::t_wave->n && p) m |= (1ull << l);
    }
    return m;
}
static inline int __any(int pred) { return __ballot(pred) != 0ull; }
static inline int __all(int pred) {
    unsigned long long full = hipemu::t_wave->n == 64 ? ~0ull : ((1ull << hipemu::t_wave->n) - 1);
    return (__ballot(pred) & full) == full;
}

// wave-level fences used by the kernels: the scheduling barrier becomes a real 64-thread rendezvous here
static inline void __builtin_amdgcn_wave_barrier() { hipemu::bar_wait(hipemu::t_wave->bar); }
static inline void __builtin_amdgcn_s_waitcnt(int) {}
static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }   // only ever applied to wave-uniform values

static inline int atomicAdd(int* p, int v) { return __sync_fetch_and_add(p, v); }
static inline float atomicAdd(float* p, float v) {
    unsigned* up = reinterpret_cast<unsigned*>(p);
    unsigned old = *up, seen;
    do {
        seen = old;
        float f;
        std::memcpy(&f, &seen, 4);
        f += v;
        unsigned nu;
        std::memcpy(&nu, &f, 4);
        old = __sync_val_compare_and_swap(up, seen, nu);
    } while (old != seen);
    float r;
    std::memcpy(&r, &old, 4);
    return r;
}
static inline unsigned __float_as_uint(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}
static inline float __uint_as_float(unsigned u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
static inline float __builtin_amdgcn_sqrtf(float x) { return std::sqrt(x); }
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
static inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }

// ---- host runtime: "device" memory is host memory ----------------------------------------------------
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
// a small "chip": persistent kernels (one workgroup per CU) then walk several items per workgroup in the tests
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 3; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s_, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t r = 0; r < h; ++r) std::memcpy((char*)d + r * dp, (const char*)s_ + r * sp, w);
    return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)std::malloc(1); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
// launches are synchronous here, so a second stream and cross-stream events order trivially
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* st, unsigned) { *st = (hipStream_t)std::malloc(1); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t st) { std::free(st); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(1); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
