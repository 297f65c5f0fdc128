#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_conv.py -m gpu -q --tb=short -k "next_norm or backbone or chunked or reloading" 2>&1 | tail -6
bash tools/ab_env.sh TAPIR_FUSE_FINALIZE 2>&1 | tail -6
