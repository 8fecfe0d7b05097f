#!/bin/bash
# usage: build_cov_variants.sh name "-Dflags" [name "-Dflags" ...]
cd "$(dirname "$0")"
while [ $# -gt 0 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -shared -fPIC $2 -o cov_bench_$1.so cov_bench.hip &
  shift 2
done
wait
ls -la cov_bench_*.so
