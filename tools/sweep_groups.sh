cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04groups; export TMPDIR=/tmp
for cfg in "4 4" "4 8" "4 12" "4 16" "2 8" "3 6" "3 12" "2 4" "4 24"; do
  set -- $cfg
  TAPIR_BACKBONE_STREAMS=$1 TAPIR_BACKBONE_GROUPS=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams $1 groups $2', d['ms_per_step'], 'bb', d['backbone_ms'], 'hot', d['hot_path_ms'])" | tee -a gpurun_out/r04groups/sweep.txt
done
