#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R=$PWD; OUT=gpurun_out/r03f; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o online -- python $R/tools/bench_online.py --frames 40 --eager-only > $R/$OUT/online.log 2> $R/$OUT/rocprof.err
cd $R; for f in $(find $OUT/prof -name '*.db'); do python profiles/summarize_rocpd.py $f > $OUT/online_kernel_stats.csv; done
find $OUT/prof -name '*.db' -size +20M -delete
head -40 $OUT/online_kernel_stats.csv | cut -c1-180
grep workload $OUT/online.log | cut -c60-200
