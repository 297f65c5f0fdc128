#!/bin/bash
# Round-5 GPU calls, one part per gpurun call; everything lands in gpurun_out/$2 (default r05).
#   tests  : pytest -m gpu (every test by name, durations), smoke
#   bench  : headline bench line with roofline_all + emulated ranks, small-N mixer A/B, f32 / BootsTAPIR / online / config 5 lines
#   prof   : rocprofv3 kernel stats of the bench command, PMC traffic + SQ passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
PART=${1:-tests}
OUT=gpurun_out/${2:-r05}
mkdir -p $OUT
export TMPDIR=/tmp
export OUT
R=$PWD
summ() { python - "$@" <<'PY'
import json,sys,os
for f in sys.argv[1:]:
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'unreadable',e); continue
    k={a: b.get('avg_us') for a, b in (d.get('kernels') or {}).items() if b.get('launches')}
    print(os.path.basename(f), d.get('ms_per_step'), d.get('value'), 'hot', d.get('hot_path_ms'), 'bb', d.get('backbone_ms'), k)
PY
}
if [ "$PART" == "tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -rA --durations=15 2>&1 | grep -E "PASSED|FAILED|ERROR|passed|failed|^[0-9.]+s " > $OUT/pytest_gpu.log; tail -22 $OUT/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
fi
if [ "$PART" == "bench" ]; then
  timeout 900 python bench.py --steps 20 --warmup 5 --emulate-rank 2 4 8 2>$OUT/bench.err | tail -1 > $OUT/bench_bf16.json; cut -c1-260 $OUT/bench_bf16.json; tail -3 $OUT/bench.err
  python - <<'PY'
import json,os
d=json.loads(open(os.environ['OUT']+'/bench_bf16.json').read())
print(json.dumps(d.get('roofline_all'),indent=0)[:3000])
print(json.dumps(d.get('emulated_ranks'),indent=0)[:3000])
print(d.get('cpu_baseline'))
PY
  KBENCH_MIXER_SHAPES=16x48,32x48,64x48,96x48,128x48 timeout 600 python tools/kbench.py --what mixer --reps 20 --out $OUT/kbench_mixer_small.json 2>&1 | grep '"kernel"' > $OUT/kbench_mixer_small.txt; cut -c1-220 $OUT/kbench_mixer_small.txt
fi
if [ "$PART" == "dbg" ]; then
  timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q -k "staged" 2>&1 | tail -60 > $OUT/dbg_staged.log; tail -40 $OUT/dbg_staged.log
  timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -x -q --durations=3 2>&1 | tail -12 > $OUT/dbg_fuzz.log; cat $OUT/dbg_fuzz.log
fi
if [ "$PART" == "c3" ]; then
  timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q -k "staged" > $OUT/dbg_staged.log 2>&1; tail -25 $OUT/dbg_staged.log
  timeout 900 python -m pytest tests/test_gpu_fuzz_parity.py -x -q --durations=3 > $OUT/dbg_fuzz.log 2>&1; tail -12 $OUT/dbg_fuzz.log
  # previous commit's library (no ring swizzle, fused mixer from 128 tracks) against the working tree; then 3 streams
  for rep in 1 2 3; do
    TAPIR_HIP_LIB=$R/tools/bin/libtapir_hip_prev.so timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_cv_prev_$rep.json
    timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_cv_new_$rep.json
    TAPIR_BACKBONE_STREAMS=3 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_cv_new_streams3_$rep.json
  done
  summ $OUT/ab_cv_*.json | tee $OUT/ab_cv_summary.txt
  timeout 300 python tools/kbench.py --what fill --reps 20 --out $OUT/kbench_fill.json 2>&1 | grep '"kernel"' > $OUT/kbench_fill.txt; cut -c1-220 $OUT/kbench_fill.txt
  timeout 400 python tools/kbench.py --what gemm --shapes costvol5,costvol5full --tiles 3,8,14,16 --reps 12 --out $OUT/kbench_cvgemm.json 2>&1 | grep '"kernel"' > $OUT/kbench_cvgemm.txt; cut -c1-330 $OUT/kbench_cvgemm.txt
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/prof_online -o online -- python $R/tools/bench_online.py --frames 30 --eager-only > $R/$OUT/online_eager.log 2> $R/$OUT/online_rocprof.err
  cd $R; for f in $(find $OUT/prof_online -name '*.db'); do python tools/online_timeline.py $f 3 > $OUT/online_timeline.txt 2>&1; done
  find $OUT/prof_online -name '*.db' -size +20M -delete
  head -40 $OUT/online_timeline.txt
fi
if [ "$PART" == "c4" ]; then
  timeout 300 python tools/probe_bf16_determinism.py > $OUT/probe_determinism.txt 2>&1; tail -12 $OUT/probe_determinism.txt
  timeout 400 python tools/kbench.py --what gemm --shapes costvol5,costvol5full --tiles 14,16 --reps 12 --out $OUT/kbench_cvgemm_plain.json 2>&1 | grep '"kernel"' > $OUT/kbench_cvgemm_plain.txt; cut -c1-330 $OUT/kbench_cvgemm_plain.txt
  TAPIR_DEBUG_GEMM_NT=1 timeout 400 python tools/kbench.py --what gemm --shapes costvol5,costvol5full --tiles 14,16 --reps 12 --out $OUT/kbench_cvgemm_nt.json 2>&1 | grep '"kernel"' > $OUT/kbench_cvgemm_nt.txt; cut -c1-330 $OUT/kbench_cvgemm_nt.txt
  timeout 300 python tools/kbench.py --what contraction --reps 20 --out $OUT/kbench_contraction.json 2>&1 | grep '"kernel"' > $OUT/kbench_contraction.txt; cut -c1-400 $OUT/kbench_contraction.txt
  TAPIR_CV_FORM=2 timeout 300 python tools/kbench.py --what contraction --reps 20 --out $OUT/kbench_contraction_form2.json 2>&1 | grep '"kernel"' > $OUT/kbench_contraction_form2.txt; cut -c1-400 $OUT/kbench_contraction_form2.txt
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16_stages.py -q -x 2>&1 | tail -5
fi
if [ "$PART" == "c5" ]; then
  timeout 300 python tools/probe_sharded_staged.py > $OUT/probe_sharded.txt 2>&1; grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_sharded.txt | tail -20
  TAPIR_HIP_LIB=$R/tools/bin/libtapir_hip_exp.so timeout 300 python tools/kbench.py --what convtrace --out $OUT/kbench_convtrace.json > $OUT/convtrace.txt 2>&1; grep -v amdgpu.ids $OUT/convtrace.txt | tail -30
  timeout 300 python tools/kbench.py --what conv --reps 20 --out $OUT/kbench_conv.json 2>&1 | grep '"kernel"' > $OUT/kbench_conv.txt; cut -c1-300 $OUT/kbench_conv.txt
fi
if [ "$PART" == "pmcbb" ]; then
  # backbone HBM traffic per clip, one launch and two launches for conv_0 + proj_conv (separate --pmc passes)
  for fp in 1 0; do
    cd /tmp
    TAPIR_FUSE_PROJ=$fp TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch_$fp -o f -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_fetch_$fp.err
    TAPIR_FUSE_PROJ=$fp TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_write_$fp -o w -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_write_$fp.err
    cd $R; python tools/pmc_traffic.py $OUT/pmc_fetch_$fp $OUT/pmc_write_$fp > $OUT/pmc_traffic_fuse_proj_$fp.json 2> $OUT/pmc_traffic_$fp.err
    python -c "import json; d=json.load(open('$OUT/pmc_traffic_fuse_proj_$fp.json')); print('FUSE_PROJ=$fp', json.dumps(d.get('backbone_per_clip')))"
    find $OUT/pmc_fetch_$fp $OUT/pmc_write_$fp -size +8M -delete
  done
fi
if [ "$PART" == "abproj" ]; then
  # conv_0 + proj_conv in one launch (TAPIR_FUSE_PROJ) on / off, alternated on the same box
  for rep in 1 2 3; do
    TAPIR_FUSE_PROJ=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_proj_off_$rep.json
    TAPIR_FUSE_PROJ=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_proj_on_$rep.json
  done
  TAPIR_BACKBONE_STREAMS=6 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_proj_on_streams6.json
  TAPIR_BACKBONE_STREAMS=3 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_proj_on_streams3.json
  summ $OUT/ab_proj_*.json | tee $OUT/ab_proj_summary.txt
fi
if [ "$PART" == "abx" ]; then
  # ExtraConvs: 64 pixels per workgroup (TAPIR_XCONV_NT=4) against the default (128 where it gives more rows), same box
  for rep in 1 2; do
    for nt in 4 0; do
      TAPIR_XCONV_NT=$nt timeout 300 python bench.py --model bootstapir --queries 1024 --steps 10 --warmup 4 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/ab_xconv_boots_nt${nt}_$rep.json
    done
  done
  for nt in 4 0; do
    TAPIR_XCONV_NT=$nt timeout 600 python tools/run_config5.py > $OUT/ab_xconv_config5_nt${nt}.json 2>/dev/null
  done
  summ $OUT/ab_xconv_boots_*.json | tee $OUT/ab_xconv_summary.txt
  cat $OUT/ab_xconv_config5_nt4.json $OUT/ab_xconv_config5_nt0.json | cut -c1-330 | tee -a $OUT/ab_xconv_summary.txt
fi
if [ "$PART" == "more" ]; then
  timeout 600 python bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_fp32.json; cut -c1-200 $OUT/bench_fp32.json
  timeout 600 python bench.py --model bootstapir --queries 1024 --no-accuracy --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_boots_q1024.json; cut -c1-200 $OUT/bench_boots_q1024.json
  timeout 300 python tools/bench_online.py --frames 60 2>&1 | grep workload > $OUT/online.json; cut -c1-230 $OUT/online.json
  timeout 600 python tools/run_config5.py > $OUT/config5_1gpu.json 2>/dev/null; cat $OUT/config5_1gpu.json
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
