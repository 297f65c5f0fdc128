set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=gpurun_out
timeout 600 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $OUT/r04_bench2.json 2> $OUT/r04_bench2.err; cut -c1-900 $OUT/r04_bench2.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_r04a -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $R/$OUT/r04_bench_under_rocprof.json 2> $R/$OUT/rocprof.err
cd $R
python profiles/summarize_rocpd.py $(ls $OUT/prof_r04a/*/*results.db $OUT/prof_r04a/*results.db 2>/dev/null | head -1) > $OUT/r04_kernel_stats_a.csv; head -40 $OUT/r04_kernel_stats_a.csv
