#!/bin/bash
# Same-box A/B of library builds: tools/ab_bench.sh A B C ...  (tools/bin/libtapir_<X>.so), two rounds each,
# alternating, bench.py --no-cpu-baseline --steps 20.  Prints ms_per_step / hot path / gemm averages per run.
cd "$(dirname "$0")/.."
cp tapnet_amd/csrc/libtapir_hip.so /tmp/libtapir_saved.so
for round in 1 2; do
  for v in "$@"; do
    cp tools/bin/libtapir_$v.so tapnet_amd/csrc/libtapir_hip.so
    python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels']
print('$v round $round: ms/step', d['ms_per_step'], 'hot', d['hot_path_ms'], 'bb', d['backbone_ms'],
      'up', k['gemm_up']['avg_us'], 'down', k['gemm_down']['avg_us'], 'mix', k['mix']['avg_us'], 'patch', k['patch_corr']['avg_us'], 'cvh', k['cv_heads']['avg_us'])"
  done
done
cp /tmp/libtapir_saved.so tapnet_amd/csrc/libtapir_hip.so
