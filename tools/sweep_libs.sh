#!/bin/bash
# same-box A/B of library builds: tools/sweep_libs.sh name1 name2 ...  (tools/bin/libtapir_hip_<name>.so; "base" = the product library)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r04libs; export TMPDIR=/tmp
for rep in 1 2; do
for name in "$@"; do
  if [ "$name" == "base" ]; then unset TAPIR_HIP_LIB; else export TAPIR_HIP_LIB=$PWD/tools/bin/libtapir_hip_$name.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={a: b.get('avg_us') for a, b in (d.get('kernels') or {}).items() if b.get('launches')}
print('$name', d['ms_per_step'], 'bb', d['backbone_ms'], 'hot', d['hot_path_ms'], k)" | tee -a gpurun_out/r04libs/sweep.txt
done; done
