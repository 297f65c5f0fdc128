#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_gpu_conv.py "tests/test_gpu_parity.py::test_backbone_golden_gpu" \
  "tests/test_gpu_parity_full.py::test_bf16_backbone_golden" "tests/test_gpu_parity.py::test_full_call_golden_with_backbone" \
  > gpurun_out/r03_pytest_d.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_pytest_d.log
grep -E "passed|failed|FAILED" gpurun_out/r03_pytest_d.log | tail -8
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-accuracy > gpurun_out/r03_bench_bf16_d$i.json 2> gpurun_out/r03_bench_d.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r03_bench_bf16_d$i.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'hot', d['hot_path_ms'], 'bb', d['backbone_ms'], 'mixer us', d['roofline']['avg_us'])
PY
done
python tools/bench_backbone.py 2>&1 | tail -5
