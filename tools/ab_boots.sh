#!/bin/bash
# like ab_bench.sh, for one rank's share of config 3 (BootsTAPIR kwargs, 1024 queries)
cd "$(dirname "$0")/.."
cp tapnet_amd/csrc/libtapir_hip.so /tmp/libtapir_saved.so
for v in "$@"; do
  cp tools/bin/libtapir_$v.so tapnet_amd/csrc/libtapir_hip.so
  python bench.py --no-cpu-baseline --model bootstapir --queries 1024 --steps 4 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernels']
print('$v boots1024: ms/step', d['ms_per_step'], 'hot', d['hot_path_ms'], 'up', k['gemm_up']['avg_us'], 'down', k['gemm_down']['avg_us'])"
done
cp /tmp/libtapir_saved.so tapnet_amd/csrc/libtapir_hip.so
