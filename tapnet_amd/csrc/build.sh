#!/bin/bash
# Builds libtapir_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
#   build.sh              the product library (tapnet_amd/csrc/libtapir_hip.so)
#   build.sh --exp        a -DTAPIR_EXPERIMENTS build for tools/kbench.py (phase traces, experiment kernels) into
#                         tools/bin/libtapir_hip_exp.so -- not next to the product library, never loaded by the package
#                         unless TAPIR_HIP_LIB points at it
#   build.sh --slp        the same sources WITH hipcc's SLP pass into tools/libtapir_hip_slp.so: the build that shows the hazard
#                         below (tools/r06.sh, TAPIR_HIP_LIB=... pytest tests/test_gpu_cotenant.py fails with it); never shipped
#
# -fno-slp-vectorize, and the check behind the build: on MI355X `v_pk_fma_f32 ... op_sel:[0,1,0]` (low result from the HIGH
# half of a source) loses its low result in lanes 48-63 while ANY other wave of the SIMD -- another stream, another process --
# issues MFMAs back to back; a dependent pair of packed FMAs does it every time (stand-alone reproducer without this
# library: tools/micro/run_cotenant_repro.py, log profiles/r06_cotenant_repro.txt).  hipcc's SLP vectoriser writes such forms
# when it packs scalar code (mix_kernel's temporal convolution: the run-to-run differences of the few-row mixer with two
# processes on one GPU, profiles/r06_cotenant_fault.txt); the hand-packed kernels (f32x2 arithmetic) never select a high half
# for a low result.  Without the pass the code object holds no packed-f32 arithmetic with an op_sel modifier at all --
# check_packed_forms.py fails the build if one ever reappears.
set -e
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -fno-slp-vectorize"
if [ "$1" == "--exp" ]; then
  shift
  mkdir -p ../../tools/bin
  exec hipcc $FLAGS -DTAPIR_EXPERIMENTS engine.hip -o ../../tools/bin/libtapir_hip_exp.so "$@"
fi
if [ "$1" == "--slp" ]; then
  shift
  exec hipcc ${FLAGS/ -fno-slp-vectorize/} engine.hip -o ../../tools/libtapir_hip_slp.so "$@"
fi
hipcc $FLAGS engine.hip -o libtapir_hip.so "$@"
python3 check_packed_forms.py libtapir_hip.so
