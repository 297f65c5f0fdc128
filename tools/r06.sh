#!/bin/bash
# Round-6 GPU calls, one part per gpurun call; everything lands in gpurun_out/$2 (default r06).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
PART=${1:-tests}
OUT=gpurun_out/${2:-r06}
mkdir -p $OUT
export TMPDIR=/tmp
export OUT
R=$PWD
# library variants of the fault hunt (profiles/r06_cotenant_fault.txt).  At the time of the hunt the product WAS the SLP build; today:
SLP=$R/tools/libtapir_hip_slp.so          # the same sources WITH hipcc's SLP pass (csrc/build.sh's flags minus -fno-slp-vectorize): the build that faults
NOSLP=$R/tapnet_amd/csrc/libtapir_hip.so  # the product (-fno-slp-vectorize)
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
if [ "$PART" == "flat" ]; then
  # the flat tiling of the 256-channel convolutions: bit identity + same-box A/B, kernel alone and whole step
  timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "flat" 2>&1 | tail -5 | tee $OUT/pytest_flat.log
  timeout 600 python tools/kbench.py --what convflat --reps 30 --out $OUT/kbench_convflat.json 2>&1 | grep '"kernel"' > $OUT/kbench_convflat.txt; cut -c1-260 $OUT/kbench_convflat.txt
  for rep in 1 2 3; do
    TAPIR_CONV_FLAT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $OUT/ab_flat_off_$rep.json
    TAPIR_CONV_FLAT=-1 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $OUT/ab_flat_auto_$rep.json
  done
  TAPIR_CONV_FLAT=1 TAPIR_BACKBONE_STREAMS=2 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $OUT/ab_flat_on_streams2.json
  TAPIR_CONV_FLAT=1 TAPIR_BACKBONE_STREAMS=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $OUT/ab_flat_on_streams1.json
  summ $OUT/ab_flat_*.json | tee $OUT/ab_flat_summary.txt
fi
if [ "$PART" == "probe" ]; then
  timeout 1500 python tools/probe_two_process.py --runs 16 > $OUT/probe_two_process.txt 2>&1; grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_two_process.txt | cut -c1-400 | tail -120
fi
if [ "$PART" == "probe2" ]; then
  # more repeats on the short stop points of the one-launch few-row GEMM path, snapshots of the first differing run dumped
  timeout 1500 python tools/probe_two_process.py --runs 150 --modes 1 --stops 1,2,3,4 --variants busy,busy+poison --skip-single --dump $OUT/probe_dumps > $OUT/probe_two_process_2.txt 2>&1
  grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_two_process_2.txt | cut -c1-420 | tail -60
fi
if [ "$PART" == "mixfault" ]; then
  export TAPIR_HIP_LIB=$SLP
  # the faulty kernel alone (mix_kernel through tapir_debug_mix): which settings produce the fault, its signature, and the
  # same probe with a build of the library without packed-f32 math (-fno-slp-vectorize)
  timeout 1200 python tools/probe_mix_fault.py --setting quiet,stream,procs-quiet,neighbour,procs --launches 6000 --tracks 64 > $OUT/probe_mix_fault.txt 2>&1
  grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_mix_fault.txt | cut -c1-260 | tail -60
  TAPIR_HIP_LIB=$NOSLP timeout 900 python tools/probe_mix_fault.py --setting procs --launches 6000 --tracks 64 > $OUT/probe_mix_fault_noslp.txt 2>&1
  grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_mix_fault_noslp.txt | cut -c1-260 | tail -30
fi
if [ "$PART" == "mixfault2" ]; then
  export TAPIR_HIP_LIB=$SLP
  # the kernel in its context (the separate-launch mixer up to launch group k, 5 tracks x 9 frames as in probe_two_process.py), fast loop
  for k in 2 3 4; do
    timeout 600 python tools/probe_mix_fault.py --setting procs --launches 8000 --tracks 5 --stop $k > $OUT/probe_ctx_stop$k.txt 2>&1
    grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_ctx_stop$k.txt | cut -c1-220 | tail -24
  done
  timeout 600 python tools/probe_mix_fault.py --setting procs --launches 8000 --tracks 5 > $OUT/probe_alone_n5.txt 2>&1
  grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_alone_n5.txt | cut -c1-220 | tail -8
  TAPIR_HIP_LIB=$NOSLP timeout 600 python tools/probe_mix_fault.py --setting procs --launches 8000 --tracks 5 --stop 4 > $OUT/probe_ctx_stop4_noslp.txt 2>&1
  grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_ctx_stop4_noslp.txt | cut -c1-220 | tail -24
  timeout 600 python tools/probe_mix_fault.py --setting stream --launches 8000 --tracks 5 --stop 4 > $OUT/probe_ctx_stop4_1proc.txt 2>&1
  grep -v "amdgpu.ids\|Gloo\|socket.cpp" $OUT/probe_ctx_stop4_1proc.txt | cut -c1-220 | tail -8
fi
if [ "$PART" == "mixfault3" ]; then
  export TAPIR_HIP_LIB=$SLP
  # who has to run beside mix_kernel for it to fault: rank 0 probes mix_kernel ALONE, rank 1 runs (unchecked) what --aggr says
  run() { timeout 600 python tools/probe_mix_fault.py --setting pair --launches 8000 --tracks 5 "$@" 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | cut -c1-230 | grep -v "^    launch" >> $OUT/probe_pairs.txt; }
  run --stop 0 --aggr 4,1,1
  run --stop 0 --aggr 4,1,0 --victim-matmuls 0
  run --stop 0 --aggr 4,0,1
  run --stop 0 --aggr 1,1,0 --victim-matmuls 0
  run --stop 0 --aggr 3,1,0 --victim-matmuls 0
  run --stop 0 --aggr 0,1,0 --victim-matmuls 0
  run --stop 4 --aggr -1,1,1
  run --stop 4 --aggr -1,1,0 --victim-matmuls 0
  cat $OUT/probe_pairs.txt
fi
if [ "$PART" == "mixfault4" ]; then
  # whose packed-f32 code matters: the victim's (mix_kernel), the aggressor's (gemm_small_kernel), or both
  run() { timeout 600 python tools/probe_mix_fault.py --setting pair --launches 8000 --tracks 5 --stop 0 --victim-matmuls 0 "$@" 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | cut -c1-230 | grep -v "^    " >> $OUT/probe_pairs_libs.txt; }
  P=$SLP; NS=$NOSLP
  run --aggr 1,1,0 --victim-lib $P --aggr-lib $P
  run --aggr 1,1,0 --victim-lib $NS --aggr-lib $P
  run --aggr 1,1,0 --victim-lib $P --aggr-lib $NS
  run --aggr 1,1,0 --victim-lib $NS --aggr-lib $NS
  for L in $EXTRA_LIBS; do run --aggr 1,1,0 --victim-lib $P --aggr-lib $L; run --aggr 1,1,0 --victim-lib $L --aggr-lib $P; done
  cat $OUT/probe_pairs_libs.txt
fi
if [ "$PART" == "mixfault5" ]; then
  export TAPIR_HIP_LIB=$SLP
  # which ingredient of the neighbouring wave: one micro kernel (tools/micro/cotenant_aggressors.hip) per run beside mix_kernel
  run() { timeout 600 python tools/probe_mix_fault.py --setting pair --launches 8000 --tracks 5 --stop 0 --victim-matmuls 0 "$@" 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | cut -c1-230 | grep -v "^    " >> $OUT/probe_pairs_micro.txt; }
  for k in 0 1 2 3 4 5 6; do run --aggr micro:$k; done
  run --aggr micro:3:256:2000
  run --aggr micro:6:32:400
  run --aggr 1,1,0
  cat $OUT/probe_pairs_micro.txt
fi
if [ "$PART" == "repro" ]; then
  # the stand-alone reproducer (no engine): packed-f32 forms next to MFMA traffic, alone / second stream / second process
  timeout 900 python tools/micro/run_cotenant_repro.py --kinds 0,3,4 > $OUT/cotenant_repro.txt 2>&1
  grep -v "amdgpu.ids" $OUT/cotenant_repro.txt | cut -c1-200
  # and the engine: mix_kernel beside the engine's own few-row GEMM built WITHOUT and with accumulation registers; the no-SLP build as victim
  run() { timeout 600 python tools/probe_mix_fault.py --setting pair --launches 8000 --tracks 5 --stop 0 --victim-matmuls 0 "$@" 2>&1 | grep -v "amdgpu.ids\|Gloo\|socket.cpp" | cut -c1-230 | grep -v "^    " >> $OUT/probe_pairs_fix.txt; }
  # (the hunt also ran an aggressor built with -mllvm -amdgpu-mfma-vgpr-form -- no AGPRs --: 58 faulty launches of 8000, i.e. it is the MFMAs)
  run --aggr micro:3 --victim-lib $SLP
  run --aggr micro:3 --victim-lib $NOSLP
  cat $OUT/probe_pairs_fix.txt
fi
if [ "$PART" == "abslp" ]; then
  # the product build (-fno-slp-vectorize) against the same sources with the SLP pass (tools/libtapir_hip_slp.so): whole step, BootsTAPIR, online
  B="python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary"
  for rep in 1 2 3; do
    TAPIR_HIP_LIB=$SLP timeout 300 $B 2>/dev/null | tail -1 > $OUT/ab_slp_on_$rep.json
    timeout 300 $B 2>/dev/null | tail -1 > $OUT/ab_slp_off_$rep.json
  done
  summ $OUT/ab_slp_*.json | tee $OUT/ab_slp_summary.txt
  TAPIR_HIP_LIB=$SLP timeout 300 python tools/bench_online.py --frames 60 2>/dev/null | grep hipGraph | grep '"auto"' | cut -c1-260 | sed 's/^/slp   /' | tee -a $OUT/ab_slp_summary.txt
  timeout 300 python tools/bench_online.py --frames 60 2>/dev/null | grep hipGraph | grep '"auto"' | cut -c1-260 | sed 's/^/noslp /' | tee -a $OUT/ab_slp_summary.txt
fi
if [ "$PART" == "fusepatch" ]; then
  timeout 300 python -m pytest tests/test_gpu_distributed.py -q -x -k "staged" 2>&1 | tail -3 | tee $OUT/pytest_staged.log
  B="python bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary"
  for rep in 1 2 3; do
    TAPIR_FUSE_PATCH=0 timeout 300 $B 2>/dev/null | tail -1 > $OUT/ab_fp_off_$rep.json
    TAPIR_FUSE_PATCH=1 timeout 300 $B 2>/dev/null | tail -1 > $OUT/ab_fp_on_$rep.json
  done
  summ $OUT/ab_fp_*.json | tee $OUT/ab_fusepatch_summary.txt
fi
if [ "$PART" == "small" ]; then
  # the few-frame form of the block convolutions: parity on the GPU, the online step with and without it, the online tests
  timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "few_frame or extra_convs_block" 2>&1 | tail -5 | tee $OUT/pytest_small.log
  for rep in 1 2; do
    TAPIR_CONV_SMALL=0 timeout 300 python tools/bench_online.py --frames 120 2>/dev/null | grep hipGraph | grep '"auto"' | cut -c1-260 | sed 's/^/small off /' | tee -a $OUT/ab_small_summary.txt
    TAPIR_CONV_SMALL=1 timeout 300 python tools/bench_online.py --frames 120 2>/dev/null | grep hipGraph | grep '"auto"' | cut -c1-260 | sed 's/^/small on  /' | tee -a $OUT/ab_small_summary.txt
  done
  timeout 900 python -m pytest tests -m gpu -q -x -k "online or causal or jax" 2>&1 | tail -5 | tee $OUT/pytest_online.log
fi
if [ "$PART" == "mlp" ]; then
  # the one-launch channel MLP of the few-row mixer (gemm.hpp mlp_small_kernel): parity, the online step with and without it, one frame launch by launch
  timeout 900 python -m pytest tests/test_gpu_bf16_stages.py -q -x -k "few_row" 2>&1 | tail -25 | tee $OUT/pytest_mlp.log
  for rep in 1 2; do
    for gm in 2 3 7 11; do
      timeout 300 python tools/bench_online.py --frames 120 --gemm-mode $gm 2>/dev/null | grep hipGraph | grep '"auto"' | cut -c1-260 | sed "s/^/gemm_mode $gm /" | tee -a $OUT/ab_mlp_summary.txt
    done
  done
  timeout 1200 python -m pytest tests -m gpu -q -x -k "online or causal or jax or distributed or sharded" 2>&1 | tail -5 | tee $OUT/pytest_online.log
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/prof_online -o online -- python $R/tools/bench_online.py --frames 30 --eager-only > $R/$OUT/online_under_trace.json 2> $R/$OUT/rocprof_online.err
  cd $R; for f in $(find $OUT/prof_online -name '*.db'); do python tools/online_timeline.py $f 3 > $OUT/online_timeline_mlp.txt 2>&1; done
  find $OUT/prof_online -name '*.db' -size +20M -delete
  head -40 $OUT/online_timeline_mlp.txt | cut -c1-150
fi
if [ "$PART" == "onl" ]; then
  # the persistent mixer of the online model: parity, phase stamps, the online step per form
  timeout 900 python -m pytest tests/test_gpu_bf16_stages.py -q -x -k "few_row or times_out" 2>&1 | tail -5 | tee $OUT/pytest_mlp.log
  timeout 300 python tools/dbg_online_tmp.py 2>&1 | grep -v amdgpu.ids | grep "True 1[01]\|False 1[01]" | tee $OUT/sync_words.txt
  for f in 0 1; do timeout 300 python tools/probe_online_mixer.py --form $f 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_online_form$f.txt; done
  for rep in 1 2; do
    for gm in 2 3 7; do
      timeout 300 python tools/bench_online.py --frames 120 --gemm-mode $gm 2>/dev/null | grep hipGraph | grep '"auto"' | cut -c1-260 | sed "s/^/gemm_mode $gm /" | tee -a $OUT/ab_mlp_summary.txt
    done
  done
fi
if [ "$PART" == "onltl" ]; then
  # one online frame launch by launch per persistent-mixer form
  for gm in 3 7; do
    cd /tmp
    timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/prof_online$gm -o online -- python $R/tools/bench_online.py --frames 30 --eager-only --gemm-mode $gm > $R/$OUT/online_under_trace$gm.json 2> $R/$OUT/rocprof_online$gm.err
    cd $R; for f in $(find $OUT/prof_online$gm -name '*.db'); do python tools/online_timeline.py $f 3 > $OUT/online_timeline_mode$gm.txt 2>&1; done
    find $OUT/prof_online$gm -name '*.db' -size +20M -delete
    head -6 $OUT/online_timeline_mode$gm.txt | cut -c1-150
  done
fi
if [ "$PART" == "onlpmc" ]; then
  # HBM traffic per launch of the online mixer's forms by the counters (separate passes, eager frames)
  for gm in 3 2; do
    cd /tmp
    timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch$gm -o f -- python $R/tools/bench_online.py --frames 6 --eager-only --gemm-mode $gm > /dev/null 2> $R/$OUT/pmc_fetch$gm.err
    timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_write$gm -o w -- python $R/tools/bench_online.py --frames 6 --eager-only --gemm-mode $gm > /dev/null 2> $R/$OUT/pmc_write$gm.err
    cd $R; python tools/pmc_traffic.py $OUT/pmc_fetch$gm $OUT/pmc_write$gm > $OUT/pmc_traffic_online_mode$gm.json 2> $OUT/pmc_traffic$gm.err
    python - <<PY
import json
d=json.load(open('$OUT/pmc_traffic_online_mode$gm.json'))
for k in ('mixer_online','mlp_small','mix_few_rows'):
    if k in d.get('kernels', d): print('mode $gm', k, json.dumps(d.get('kernels', d)[k])[:400])
PY
    find $OUT/pmc_fetch$gm $OUT/pmc_write$gm -size +8M -delete
  done
fi
if [ "$PART" == "warmside" ]; then
  # the weight warm-up of a clip's first refinement iteration on a side stream under the cost volume vs in front of the mixer
  timeout 900 python -m pytest tests -m gpu -q -x -k "headline or full_call or hot_path or fuzz or bench_launch" 2>&1 | tail -3 | tee $OUT/pytest_warm.log
  for rep in 1 2 3; do
    for v in 0 1; do
      TAPIR_WARM_SIDE=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $OUT/ab_warm_${v}_$rep.json
    done
  done
  summ $OUT/ab_warm_*.json | tee $OUT/ab_warm_summary.txt
fi
if [ "$PART" == "iter0" ]; then
  # iter0's copies written by the cost-volume kernel vs as a launch of their own
  timeout 1200 python -m pytest tests -m gpu -q -x -k "headline or full_call or hot_path or fuzz or cost_volume or golden or bench_launch" 2>&1 | tail -3 | tee $OUT/pytest_iter0.log
  for rep in 1 2 3; do
    for v in 0 1; do
      TAPIR_FUSE_ITER0=$v timeout 300 python bench.py --steps 30 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 > $OUT/ab_iter0_${v}_$rep.json
    done
  done
  summ $OUT/ab_iter0_*.json | tee $OUT/ab_iter0_summary.txt
fi
if [ "$PART" == "onlinetl" ]; then
  # one online frame launch by launch, with and without the few-frame convolutions
  for m in 1 0; do
    cd /tmp
    TAPIR_CONV_SMALL=$m timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/prof_online$m -o online -- python $R/tools/bench_online.py --frames 30 --eager-only > $R/$OUT/online_under_trace$m.json 2> $R/$OUT/rocprof_online$m.err
    cd $R; for f in $(find $OUT/prof_online$m -name '*.db'); do python tools/online_timeline.py $f 3 > $OUT/online_timeline_small$m.txt 2>&1; done
    find $OUT/prof_online$m -name '*.db' -size +20M -delete
    head -40 $OUT/online_timeline_small$m.txt | cut -c1-150
  done
fi
if [ "$PART" == "timeline" ]; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/prof_tl -o bench -- python $R/bench.py --steps 12 --warmup 4 --no-accuracy --no-cpu-baseline --no-secondary > $R/$OUT/bench_under_trace.json 2> $R/$OUT/rocprof_tl.err
  cd $R; for f in $(find $OUT/prof_tl -name '*.db'); do python tools/timeline.py $f 3 -v > $OUT/step_timeline.txt 2>&1; done
  find $OUT/prof_tl -name '*.db' -size +20M -delete
  grep -v "conv_fused\|stem_conv" $OUT/step_timeline.txt | head -70
fi
if [ "$PART" == "flatk" ]; then
  timeout 600 python -m pytest tests/test_gpu_conv.py -q -x -k "flat" 2>&1 | tail -5 | tee $OUT/pytest_flat.log
  timeout 600 python tools/kbench.py --what convflat --reps 30 --out $OUT/kbench_convflat.json 2>&1 | grep '"kernel"' > $OUT/kbench_convflat.txt; cut -c1-260 $OUT/kbench_convflat.txt | grep -v "rep\": 1"
  TAPIR_HIP_LIB=$R/tools/libtapir_hip_exp.so timeout 300 python tools/kbench.py --what convflattrace,convtrace --out $OUT/kbench_convtrace.json > $OUT/convflattrace.txt 2>&1; grep -v amdgpu.ids $OUT/convflattrace.txt | tail -50
fi
if [ "$PART" == "bench" ]; then
  timeout 1200 python bench.py --steps 25 --warmup 5 2>$OUT/bench.err | tail -1 > $OUT/bench_bf16.json; cut -c1-260 $OUT/bench_bf16.json; tail -3 $OUT/bench.err
  python - <<'PY'
import json,os
d=json.loads(open(os.environ['OUT']+'/bench_bf16.json').read())
print(json.dumps(d.get('secondary'),indent=0)[:3000])
print(json.dumps(d.get('roofline_all'),indent=0)[:3000])
print(d.get('cpu_baseline'))
PY
fi
if [ "$PART" == "prof" ]; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline --no-secondary > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
  cd $R; for f in $(find $OUT/prof -name '*.db'); do python profiles/summarize_rocpd.py $f > $OUT/kernel_stats.csv; done
  find $OUT/prof -name '*.db' -size +20M -delete
  head -24 $OUT/kernel_stats.csv | cut -c1-150
  cd /tmp
  TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline --no-secondary > /dev/null 2> $R/$OUT/pmc_fetch.err
  TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline --no-secondary > /dev/null 2> $R/$OUT/pmc_write.err
  TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/$OUT/pmc_sq -o q -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline --no-secondary > /dev/null 2> $R/$OUT/pmc_sq.err
  cd $R; python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err; head -40 $OUT/pmc_traffic.json
  python tools/pmc_sq.py $OUT/pmc_sq > $OUT/pmc_sq.txt 2>$OUT/pmc_sq.err; head -30 $OUT/pmc_sq.txt
  find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq -size +8M -delete
fi
