#!/bin/bash
# round-3 GPU call E: cost-volume XCD map A/B + tests; rocprofv3 kernel stats + PMC passes of the bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
R=$PWD; OUT=gpurun_out/r03e; mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest -q -m gpu tests/test_gpu_parity.py -k "cost_volume or tapnet or config2 or hot_path" tests/test_gpu_bf16_stages.py::test_cv_fused_bf16_vs_oracle 2>&1 | tail -3
for lib in base new base new; do
  if [ $lib = base ]; then export TAPIR_HIP_LIB=$R/tapnet_amd/csrc/libtapir_hip_base.so; else unset TAPIR_HIP_LIB; fi
  echo "== $lib"; python tools/kbench.py --what cv --reps 40 --out $OUT/kbench_cv_$lib.json 2>&1 | grep -E "fused|kernel" | cut -c1-220
done
unset TAPIR_HIP_LIB
for m in hip torch; do echo "online extra_convs=$m"; TAPIR_EXTRA_CONVS=$m timeout 300 python tools/bench_online.py --frames 40 2>&1 | grep workload | cut -c60-200; done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-accuracy --no-cpu-baseline > $R/$OUT/bench_under_rocprof.json 2> $R/$OUT/rocprof.err
cd $R; for f in $(find $OUT/prof -name '*.db'); do python profiles/summarize_rocpd.py $f > $OUT/kernel_stats.csv; done
find $OUT/prof -name '*.db' -size +20M -delete
head -12 $OUT/kernel_stats.csv | cut -c1-150
cd /tmp
TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_fetch.err
TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/$OUT/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_write.err
TAPIR_BACKBONE_GRAPH=0 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/$OUT/pmc_sq -o q -- python $R/bench.py --steps 3 --warmup 1 --no-accuracy --no-cpu-baseline > /dev/null 2> $R/$OUT/pmc_sq.err
cd $R; python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err
python tools/pmc_sq.py $OUT/pmc_sq > $OUT/pmc_sq.txt 2> $OUT/pmc_sq.err2; cat $OUT/pmc_sq.txt | cut -c1-170
python -c "
import json; d=json.load(open('$OUT/pmc_traffic.json'))['kernels']
for k,v in d.items(): print(k, v['fetch_bytes'], v['write_bytes'], v['launches'])"
find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq -size +8M -delete
