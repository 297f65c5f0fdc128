#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest -q -m gpu tests/test_gpu_conv.py \
  "tests/test_gpu_parity.py::test_backbone_golden_gpu" "tests/test_gpu_parity.py::test_full_call_golden_with_backbone" \
  "tests/test_gpu_parity_full.py" tests/test_gpu_bf16_stages.py tests/test_gpu_aj_proxy.py -s \
  > gpurun_out/r03_pytest_c.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03_pytest_c.log
grep -E "passed|failed|min cosine|FAILED|full call" gpurun_out/r03_pytest_c.log | tail -12
