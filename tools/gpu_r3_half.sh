#!/bin/bash
# half-CU mixer kernel: start skew on / off, same box
for div in ${DIVS:-256 0}; do
  echo "== TAPIR_HALF_SKEW_DIV=$div"
  TAPIR_HALF_SKEW_DIV=$div KBENCH_MIXER_SHAPES=${SHAPES:-1024x48,512x48} python tools/kbench.py --what mixer --dtype bfloat16 --out gpurun_out/kbench_half_$div.json 2>&1 | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
  r = json.loads(l)
  if 'separate' not in r['kernel']: print(r['kernel'][20:], r['N'], r['T'], r['med_us'], r['tflops'], r.get('max_abs_diff_vs_separate'))
"
done
