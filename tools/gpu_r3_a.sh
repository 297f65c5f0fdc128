#!/bin/bash
# round-3 GPU call A: the new parity tests, the bench line in both builds, a kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest -q -m gpu tests/test_gpu_bf16_stages.py tests/test_gpu_aj_proxy.py \
  tests/test_gpu_bench_launch.py tests/test_gpu_parity_full.py::test_bf16_end_to_end_vs_oracle \
  tests/test_gpu_distributed.py -s > gpurun_out/r03_pytest_new.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_pytest_new.log
tail -5 gpurun_out/r03_pytest_new.log
true
true
true
true
