// Launches of the persistent room pass k_room_cov_dma<M, K, SUB> (k_room.h): api_room_s8.hip instantiates the shapes (SUB = 8 time
// sub-chunks per workgroup; SUB = 4 was measured equally fast and less accurate -- profiles/r05_c5_accumulation.txt -- and removed).
#pragma once
#include "host.h"
#include "k_room.h"

namespace disco_host {
using namespace disco;

// (M, K) shapes of the one-pass room covariance (wide shapes: P = M + K - 1 > 8)
#define DISCO_FOR_ROOM(X_) X_(8, 8) X_(8, 6) X_(8, 4) X_(8, 2) X_(4, 8) X_(4, 6)

bool launch_room_s8(int M, int K, unsigned nwg, hipStream_t st, const RoomArgs& a);

#define DISCO_DEFINE_ROOM_LAUNCHER(NAME_, SUB_)                                                                             \
    bool NAME_(int M, int K, unsigned nwg, hipStream_t st, const RoomArgs& a) {                                            \
        DISCO_FOR_ROOM(DISCO_ROOM_CASE_##SUB_)                                                                              \
        return false;                                                                                                       \
    }
#define DISCO_ROOM_CASE_(M_, K_, SUB_)                                                                                      \
    if (M == M_ && K == K_) {                                                                                               \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_room_cov_dma<M_, K_, SUB_>), dim3(nwg), dim3(RoomGeomS<M_, K_, SUB_>::NT), 0, st, a); \
        return true;                                                                                                        \
    }
#define DISCO_ROOM_CASE_8(M_, K_) DISCO_ROOM_CASE_(M_, K_, 8)
}  // namespace disco_host
