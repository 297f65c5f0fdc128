#!/bin/bash
# same-box A/B of an environment switch: tools/ab_env.sh NAME v1 v2 [reps]   (bench.py quick line per value, alternated)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp; mkdir -p gpurun_out/ab_env
for rep in $(seq 1 ${4:-2}); do for v in $2 $3; do
  env $1=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={a: b.get('avg_us') for a, b in (d.get('kernels') or {}).items() if b.get('launches')}
print('$1=$v', d['ms_per_step'], 'bb', d['backbone_ms'], 'hot', d['hot_path_ms'], k)" | tee -a gpurun_out/ab_env/$1.txt
done; done
