#!/bin/bash
# same-box A/B of library builds: bash tools/ab_lib.sh <lib1.so> <lib2.so> ... [-- bench args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in 1 2; do
for lib in "${LIBS[@]}"; do
  TAPIR_HIP_LIB=$PWD/tapnet_amd/csrc/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$lib', d['value'], d['ms_per_step'], 'hot', d['hot_path_ms'], 'patch us', k['patch_corr']['avg_us'], 'cv us', k['cv_heads']['avg_us'])"
done
done
