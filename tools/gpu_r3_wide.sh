#!/bin/bash
# wide mixer kernel: explicit LDS addressing vs the committed version, same box (kbench + BootsTAPIR Q=1024 bench)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for lib in libtapir_hip_base.so libtapir_hip.so; do
  echo "== $lib"
  TAPIR_HIP_LIB=$PWD/tapnet_amd/csrc/$lib KBENCH_MIXER_SHAPES=1024x48,512x48,256x96 python tools/kbench.py --what mixer --dtype bfloat16 --out gpurun_out/kb_$lib.json 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
  r = json.loads(l)
  if 'wide' in r['kernel']: print(r['kernel'][20:], r['N'], r['T'], r['med_us'], r['tflops'], r.get('max_abs_diff_vs_separate'))
"
done
done
for lib in libtapir_hip_base.so libtapir_hip.so; do
  TAPIR_HIP_LIB=$PWD/tapnet_amd/csrc/$lib python bench.py --model bootstapir --queries 1024 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib boots q1024', d['ms_per_step'], 'hot', d.get('hot_path_ms'), 'bb', d.get('backbone_ms'))"
done
