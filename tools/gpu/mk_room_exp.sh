#!/bin/bash
# Timing-only variants of the room pass (k_room.h: DISCO_ROOM_EXP) as exp_libs/libdisco_roomexp<N>.so: only api_room_s8.hip is recompiled, the
# other objects are the default build's.  Run HERE (no GPU needed), then `gpurun -- bash tools/gpu/r4_u.sh`.   Usage: mk_room_exp.sh 1 2 4 8 3 7
set -e
cd "$(dirname "$0")/../.."
mkdir -p exp_libs
for e in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -DDISCO_ROOM_EXP=$e -c -o /tmp/room_s8_exp$e.o disco_amd/csrc/api_room_s8.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp_libs/libdisco_roomexp$e.so $(ls disco_amd/lib/obj/*.default.o | grep -v api_room_s8) /tmp/room_s8_exp$e.o
  echo "built exp_libs/libdisco_roomexp$e.so"
done
