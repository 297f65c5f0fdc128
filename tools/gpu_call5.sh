set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_bf16_stages.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/r04_t5.log 2>&1; tail -5 $OUT/r04_t5.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $OUT/r04_bench3.json 2> $OUT/r04_bench3.err; cut -c1-300 $OUT/r04_bench3.json; python -c "
import json;d=json.load(open('$OUT/r04_bench3.json'));print(d['ms_per_step'],d['hot_path_ms'],d['backbone_ms'],d['kernels'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_r04b -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $R/$OUT/r04_bench_under_rocprof.json 2> $R/$OUT/rocprof.err
cd $R
python profiles/summarize_rocpd.py $(ls $OUT/prof_r04b/*/*results.db $OUT/prof_r04b/*results.db 2>/dev/null | head -1) > $OUT/r04_kernel_stats_b.csv; head -32 $OUT/r04_kernel_stats_b.csv | cut -c1-150
