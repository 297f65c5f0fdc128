set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16_stages.py -m gpu -x -q > $OUT/r04_t8.log 2>&1; tail -3 $OUT/r04_t8.log
for fp in 0 1 0 1; do
TAPIR_FUSE_PATCH=$fp timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('fuse_patch=$fp', d['ms_per_step'],d['hot_path_ms'],d['backbone_ms'],{k:v['avg_us'] for k,v in d['kernels'].items() if v['launches']})"
done
for st in 3 6; do
TAPIR_BACKBONE_STREAMS=$st timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('streams=$st', d['ms_per_step'],d['hot_path_ms'],d['backbone_ms'])"
done
