cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rows_of_up_to_64 or resolution_512" 2>&1 | tail -12
