#!/bin/bash
# same-box A/B of an engine switch read from the environment at tapir_create (TAPIR_FUSE_UPDATE, TAPIR_SMALL_GEMM):
#   bash tools/ab_env.sh TAPIR_FUSE_UPDATE [bench.py args]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
VAR=$1; shift
for rep in 1 2 3; do
for v in 0 1; do
  env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], d['ms_per_step'], 'hot', d['hot_path_ms'], 'bb', d['backbone_ms'], 'dominant kernel us', d['roofline']['avg_us'])"
done
done
