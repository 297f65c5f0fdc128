#!/bin/bash
# same-box A/B of two builds of the library on the backbone (and the whole bench): base vs new, interleaved
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R=$PWD
for rep in 1 2; do
for lib in base new; do
  if [ $lib = base ]; then export TAPIR_HIP_LIB=$R/tapnet_amd/csrc/libtapir_hip_base.so; else unset TAPIR_HIP_LIB; fi
  echo "== $lib"
  python tools/bench_backbone.py --reps 30 2>&1 | grep -E '"streams": (1|4), "frames_per_group": null' | cut -c40-200
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], 'hot', d['hot_path_ms'], 'bb', d['backbone_ms'])"
done
done
