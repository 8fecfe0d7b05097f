#!/bin/bash
# SURVEY section 5 "race detection / sanitizers": the kernel sources of disco_amd/csrc, compiled by g++ for the hipemu test emulator with
# AddressSanitizer + UndefinedBehaviorSanitizer, driven through the emulated kernel suites.  What this leg can see: out-of-bounds global /
# LDS / stack accesses of the kernels and of the host side of the C ABI, use of freed blocks, signed overflow, misaligned or null
# dereferences, out-of-range shifts -- NOT data races (the emulator runs a workgroup's threads as fibers of one OS thread).
# The fibers switch stacks with swapcontext, which ASan only follows with annotations the emulator does not make: detect_stack_use_after_return
# stays off and fake stacks are not used.   Usage: bash tests/run_sanitized.sh [pytest args]      (log: profiles/r05_sanitizer_run.log)
cd "$(dirname "$0")/.."
export DISCO_CXXFLAGS="-fsanitize=address,undefined -fno-omit-frame-pointer -fno-sanitize-recover=undefined -g1"
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1:halt_on_error=1"
export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1"
python tests/emu_build.py || exit 1
exec python -m pytest tests/test_kernels_emulated.py tests/test_kernels_emulated_wide.py tests/test_reference_surface_emulated.py -q -x -p no:cacheprovider "$@"
