#!/bin/bash
# Round-end artefacts on one GPU box: pytest -m gpu, smoke, bench, rocprofv3 kernel stats of the bench
# command, two PMC passes (FETCH_SIZE / WRITE_SIZE) of a short bench.  Everything lands in gpurun_out/$1.
OUT=gpurun_out/${1:-final}
mkdir -p $OUT
R=$PWD
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > $OUT/bench.json; cut -c1-300 $OUT/bench.json
timeout 300 python bench.py --model bootstapir --queries 1024 --no-accuracy --no-cpu-baseline 2>&1 | tail -1 > $OUT/bench_boots_q1024.json; cut -c1-200 $OUT/bench_boots_q1024.json
if [ "$2" != "noprof" ]; then
  export TMPDIR=/tmp; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
  cd $R; for f in $(find $OUT/prof -name '*.db'); do python profiles/summarize_rocpd.py $f > $OUT/kernel_stats.csv; done
  find $OUT/prof -name '*.db' -size +20M -delete
  head -16 $OUT/kernel_stats.csv | cut -c1-150
  cd /tmp
  TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_fetch.err
  TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_write.err
  cd $R; python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err; head -30 $OUT/pmc_traffic.json
  find $OUT/pmc_fetch $OUT/pmc_write -size +8M -delete
fi
