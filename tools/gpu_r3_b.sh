#!/bin/bash
# round-3 GPU call B: ExtraConvs in HIP -- tests, BootsTAPIR bench A/B, kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_conv.py -k "extra_convs" \
  "tests/test_gpu_parity.py::test_backbone_golden_gpu" "tests/test_gpu_parity.py::test_full_call_golden_with_backbone" \
  "tests/test_gpu_parity_full.py::test_bf16_backbone_golden" tests/test_gpu_bf16_stages.py tests/test_gpu_aj_proxy.py -s \
  > gpurun_out/r03_pytest_b.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_pytest_b.log
grep -E "passed|failed|min cosine|FAILED" gpurun_out/r03_pytest_b.log | tail -12
for mode in hip torch; do
  TAPIR_EXTRA_CONVS=$mode timeout 600 python bench.py --model bootstapir --queries 1024 --steps 10 --warmup 3 \
    --no-cpu-baseline --no-accuracy > gpurun_out/r03_bench_bootstapir_q1024_$mode.json 2> gpurun_out/r03_bench_boots_$mode.err
  python - <<PY
import json
try:
  d=json.loads(open('gpurun_out/r03_bench_bootstapir_q1024_$mode.json').read().strip().splitlines()[-1])
  print('$mode', d['value'], d['ms_per_step'], 'hot', d['hot_path_ms'], 'bb', d['backbone_ms'])
except Exception as e:
  print('$mode failed', e); print(open('gpurun_out/r03_bench_boots_$mode.err').read()[-1500:])
PY
done
cd /tmp && TAPIR_BACKBONE_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_boots -o boots -- \
  python $GRAFT_REPO_ROOT/bench.py --model bootstapir --queries 1024 --steps 5 --warmup 2 --no-cpu-baseline --no-accuracy \
  > $GRAFT_REPO_ROOT/gpurun_out/r03_bench_boots_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r03_rocprof_boots.err
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_boots | head; find gpurun_out/prof_boots -name "*kernel_stats*" | head -3
for f in $(find gpurun_out/prof_boots -name '*.db'); do python profiles/summarize_rocpd.py $f > gpurun_out/r03_boots_kernel_stats.csv; done
find gpurun_out/prof_boots -name '*.db' -size +20M -delete
head -14 gpurun_out/r03_boots_kernel_stats.csv | cut -c1-160
