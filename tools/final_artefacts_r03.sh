#!/bin/bash
# Round-3 artefacts on one GPU box: pytest -m gpu (verbose names), smoke, bench lines (bf16 headline, f32 build,
# BootsTAPIR Q=1024), the online step.  Everything lands in gpurun_out/$1.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-r03final}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed" > $OUT/pytest_gpu.log; tail -2 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_bf16.json; cut -c1-260 $OUT/bench_bf16.json
timeout 600 python bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_fp32.json; cut -c1-200 $OUT/bench_fp32.json
timeout 600 python bench.py --model bootstapir --queries 1024 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_boots_q1024.json; cut -c1-200 $OUT/bench_boots_q1024.json
timeout 300 python tools/bench_online.py --frames 60 2>&1 | grep workload > $OUT/online.json; cat $OUT/online.json | cut -c1-230
