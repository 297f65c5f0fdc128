#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 900 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "causal or online or update_query or track_many or ragged" 2>&1 | tail -3
for gm in 0 1 0 1; do timeout 300 python tools/bench_online.py --frames 60 --gemm-mode $gm 2>&1 | grep workload | head -2 | cut -c60-230; done
