#!/bin/bash
# Builds libtapir_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result \
  engine.hip -o libtapir_hip.so "$@"
