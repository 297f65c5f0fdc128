#!/bin/bash
# Round-4 artefacts on one GPU box, in three parts (a GPU call each):
#   tests : pytest -m gpu (every test by name), smoke
#   bench : bench lines (bf16 headline with the live reference CPU leg, f32 build, BootsTAPIR Q=1024, batch, online,
#           config 5), kernel micro-benchmarks
#   prof  : rocprofv3 kernel stats of the bench command, PMC traffic + SQ passes
# Everything lands in gpurun_out/$2 (default r04final).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
PART=${1:-tests}
OUT=gpurun_out/${2:-r04final}
mkdir -p $OUT
export TMPDIR=/tmp
export OUT
R=$PWD
if [ "$PART" == "tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -rA 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed" > $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
fi
if [ "$PART" == "bench" ]; then
  timeout 900 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_bf16.json; cut -c1-260 $OUT/bench_bf16.json
  timeout 600 python bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_fp32.json; cut -c1-200 $OUT/bench_fp32.json
  timeout 600 python bench.py --model bootstapir --queries 1024 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_boots_q1024.json; cut -c1-200 $OUT/bench_boots_q1024.json
  timeout 300 python tools/bench_online.py --frames 60 2>&1 | grep workload > $OUT/online.json; cut -c1-230 $OUT/online.json
  timeout 600 python tools/run_config5.py > $OUT/config5_1gpu.json 2>/dev/null; cat $OUT/config5_1gpu.json
  timeout 300 python tools/kbench.py --what cv,contraction --reps 20 --out $OUT/kbench_cv.json 2>&1 | grep '"kernel"' > $OUT/kbench_cv.txt; cut -c1-200 $OUT/kbench_cv.txt
fi
if [ "$PART" == "ab" ]; then
  # same-box A/B of the last commit's library (tools/bin/libtapir_hip_prev.so, built from `git show HEAD:...`) with the
  # identity resize still executed, against the working tree; alternated twice
  for rep in 1 2; do
    TAPIR_HIP_LIB=$R/tools/bin/libtapir_hip_prev.so TAPIR_IDENTITY_RESIZE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_prev_$rep.json
    timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_new_$rep.json
  done
  TAPIR_IDENTITY_RESIZE=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_new_with_resize.json
  python - <<'EOF' | tee $OUT/ab_summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ.get('OUT','gpurun_out/r04final')+'/ab_*.json')):
    try: d=json.loads(open(f).read())
    except Exception as e: print(f,'unreadable',e); continue
    k={a: b.get('avg_us') for a, b in (d.get('kernels') or {}).items() if b.get('launches')}
    print(os.path.basename(f), d.get('ms_per_step'), d.get('value'), 'hot', d.get('hot_path_ms'), 'bb', d.get('backbone_ms'), k)
EOF
fi
if [ "$PART" == "warm" ]; then
  # the weight-stream warming in front of a clip's first refinement iteration, on / off, alternated; then the timeline
  for rep in 1 2; do
    TAPIR_WARM_WEIGHTS=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_cold_$rep.json
    TAPIR_WARM_WEIGHTS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_warm_$rep.json
  done
  python - <<'EOF' | tee $OUT/ab_summary.txt
import json,glob,os
for f in sorted(glob.glob(os.environ.get('OUT','gpurun_out/r04final')+'/ab_*.json')):
    try: d=json.loads(open(f).read())
    except Exception as e: print(f,'unreadable',e); continue
    k={a: b.get('avg_us') for a, b in (d.get('kernels') or {}).items() if b.get('launches')}
    print(os.path.basename(f), d.get('ms_per_step'), d.get('value'), 'hot', d.get('hot_path_ms'), 'bb', d.get('backbone_ms'), k)
EOF
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
  cd $R; for f in $(find $OUT/prof -name '*.db'); do python tools/timeline.py $f 3 > $OUT/timeline.txt; python profiles/summarize_rocpd.py $f > $OUT/kernel_stats.csv; done
  cat $OUT/timeline.txt | tail -24
fi
if [ "$PART" == "prof" ]; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
  cd $R; for f in $(find $OUT/prof -name '*.db'); do python profiles/summarize_rocpd.py $f > $OUT/kernel_stats.csv; done
  find $OUT/prof -name '*.db' -size +20M -delete
  head -24 $OUT/kernel_stats.csv | cut -c1-150
  cd /tmp
  TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_fetch.err
  TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_write.err
  TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/$OUT/pmc_sq -o q -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_sq.err
  cd $R; python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err; head -40 $OUT/pmc_traffic.json
  python tools/pmc_sq.py $OUT/pmc_sq > $OUT/pmc_sq.txt 2>$OUT/pmc_sq.err; head -30 $OUT/pmc_sq.txt
  find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq -size +8M -delete
fi
