#!/bin/bash
# A/B library for one GPU pass: the named units recompiled with extra -D flags, every other object taken from the default build
# (disco_amd/lib/obj/*.default.o, which must be current: run `python -c "from disco_amd import build; build.build_hip()"` first).
# Run HERE (hipcc cross-compiles), the library travels to the GPU box in exp_libs/.
#   usage: mk_variant.sh <name> "<flags>" unit [unit ...]        e.g.  mk_variant.sh pf3 "-DDISCO_PF_DIST=3" api_stft_cov api_stft
set -e
cd "$(dirname "$0")/../.."
name=$1; flags=$2; shift 2
mkdir -p exp_libs /tmp/variant_$name
objs=$(ls disco_amd/lib/obj/*.default.o)
for u in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC $flags -c -o /tmp/variant_$name/$u.o disco_amd/csrc/$u.hip \
      -Rpass-analysis=kernel-resource-usage 2> /tmp/variant_$name/$u.remarks &
  objs=$(echo "$objs" | grep -v "/$u.default.o")
done
wait
for u in "$@"; do test -s /tmp/variant_$name/$u.o || { echo "compile of $u failed"; tail -20 /tmp/variant_$name/$u.remarks; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp_libs/libdisco_$name.so $objs $(for u in "$@"; do echo /tmp/variant_$name/$u.o; done)
echo "built exp_libs/libdisco_$name.so ($flags)"
