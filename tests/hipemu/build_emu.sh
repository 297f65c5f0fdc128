#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the unmodified HIP sources for the host
# against tests/hipemu/hip/hip_runtime.h -> tests/hipemu/libtapir_emu.so
#   build_emu.sh [flags]         the emulator the CPU tests load
#   build_emu.sh --asan [flags]  the same with -fsanitize=address,undefined -> libtapir_emu_asan.so: every emulated kernel
#                                test becomes an out-of-bounds test of the global-memory accesses (device buffers are
#                                heap allocations with red zones).  Run through tests/hipemu/run_asan.sh.
set -e
cd "$(dirname "$0")"
OUT=libtapir_emu.so
SAN=""
if [ "$1" == "--asan" ]; then
  shift
  OUT=libtapir_emu_asan.so
  SAN="-fsanitize=address,undefined -fno-sanitize=vptr,function -fno-omit-frame-pointer -shared-libsan -g1"
fi
/opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O2 -fopenmp -fPIC -shared -I. -DTAPIR_EXPERIMENTS $SAN \
  -Wno-unused-value ../../tapnet_amd/csrc/engine.hip emu_switch.cpp -o $OUT "$@"
