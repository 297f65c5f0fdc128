#!/bin/bash
# TEST INFRASTRUCTURE ONLY: runs emulated kernel tests against the AddressSanitizer / UBSan build of the emulator
# (build_emu.sh --asan).  The sanitizer runtime has to be loaded before python's own allocator:
#   tests/hipemu/run_asan.sh tests/test_kernels_emulated.py -k cost_volume -x -q
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
[ -f "$HERE/libtapir_emu_asan.so" ] || "$HERE/build_emu.sh" --asan -O1
export TAPIR_EMU_LIB="$HERE/libtapir_emu_asan.so"
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD="$RT" exec python -m pytest -p no:cacheprovider -m "not gpu" "$@"
