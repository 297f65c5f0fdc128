#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_parity.py tests/test_gpu_parity_full.py tests/test_gpu_bf16_stages.py 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], 'hot', d['hot_path_ms'], 'bb', d['backbone_ms'], 'mixer', d['roofline']['avg_us'])"
done
