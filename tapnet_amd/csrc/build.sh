#!/bin/bash
# Builds libtapir_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   build.sh              the product library (tapnet_amd/csrc/libtapir_hip.so)
#   build.sh --exp        a -DTAPIR_EXPERIMENTS build for tools/kbench.py (phase traces, experiment kernels) into
#                         tools/bin/libtapir_hip_exp.so -- not next to the product library, never loaded by the package
#                         unless TAPIR_HIP_LIB points at it
set -e
cd "$(dirname "$0")"
if [ "$1" == "--exp" ]; then
  shift
  mkdir -p ../../tools/bin
  exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -DTAPIR_EXPERIMENTS \
    engine.hip -o ../../tools/bin/libtapir_hip_exp.so "$@"
fi
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result \
  engine.hip -o libtapir_hip.so "$@"
