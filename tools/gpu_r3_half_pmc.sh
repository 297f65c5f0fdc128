#!/bin/bash
# SQ counters + L2 hit / fetch of the three track-resident mixer kernels at 512 tracks x 48 frames
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=gpurun_out/half_pmc; mkdir -p $R/$OUT
CMD="python $R/tools/kbench.py --what mixer --dtype bfloat16 --reps 8 --out $R/$OUT/kb.json"
export KBENCH_MIXER_SHAPES=512x48
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $R/$OUT/pmc_sq -o q -- $CMD > /dev/null 2> $R/$OUT/pmc_sq.err
timeout 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $R/$OUT/pmc_l2 -o f -- $CMD > /dev/null 2> $R/$OUT/pmc_l2.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $R/$OUT/pmc_sq2 -o q2 -- $CMD > /dev/null 2> $R/$OUT/pmc_sq2.err
cd $R
python tools/pmc_sq.py $OUT/pmc_sq | cut -c1-170 | tee $OUT/pmc_sq.txt
python - <<'P'
import sys, collections
sys.path.insert(0, 'tools')
from pmc_traffic import rows_of
for d in ('gpurun_out/half_pmc/pmc_l2', 'gpurun_out/half_pmc/pmc_sq2'):
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  try:
    for name, cname, value in rows_of(d):
      for key in ('mixer_fused_kernel', 'mixer_fused_wide_kernel', 'mixer_fused_half_kernel'):
        if key in name: acc[key][cname].append(value)
  except Exception as e:
    print(d, 'failed', e); continue
  for k, c in acc.items():
    print(d.split('/')[-1], k, {n: round(sum(v) / len(v)) for n, v in c.items()})
P
find $OUT -size +8M -delete
