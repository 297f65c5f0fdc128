#!/bin/bash
# TEST INFRASTRUCTURE ONLY: compiles the unmodified HIP sources for the host
# against tests/hipemu/hip/hip_runtime.h -> tests/hipemu/libtapir_emu.so
set -e
cd "$(dirname "$0")"
/opt/rocm/lib/llvm/bin/clang++ -x c++ -std=c++17 -O2 -fopenmp -fPIC -shared -I. -DTAPIR_EXPERIMENTS \
  -Wno-unused-value ../../tapnet_amd/csrc/engine.hip emu_switch.cpp -o libtapir_emu.so "$@"
