set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out
timeout 1200 python -m pytest tests/test_gpu_fuzz_parity.py tests/test_gpu_parity_full.py -m gpu -x -q -s > $OUT/r04_t6.log 2>&1; grep -E "headline_|sweep|margin mask|passed|failed|Error" $OUT/r04_t6.log | cut -c1-400
cat $OUT/parity_full_mask.txt
