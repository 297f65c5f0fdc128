#!/bin/bash
# rocprofv3 kernel stats of the bench command on the round's last build (graph replay off so that every kernel is traced)
cd "$GRAFT_REPO_ROOT"; R=$PWD; OUT=gpurun_out/r03stats; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
cd $R; for f in $(find $OUT/prof -name '*.db'); do python profiles/summarize_rocpd.py $f > $OUT/kernel_stats.csv; done
find $OUT/prof -name '*.db' -size +20M -delete
head -14 $OUT/kernel_stats.csv | cut -c1-160
tail -1 $OUT/bench_under_rocprof.json | cut -c1-200
