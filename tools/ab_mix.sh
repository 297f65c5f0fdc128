#!/bin/bash
# same-box A/B of the mix kernel alone: tools/ab_mix.sh old new
cd "$(dirname "$0")/.."
cp tapnet_amd/csrc/libtapir_hip.so /tmp/libtapir_saved.so
for v in "$@" "$@"; do
  cp tools/bin/libtapir_$v.so tapnet_amd/csrc/libtapir_hip.so
  python tools/kbench.py --what mix --reps 30 2>/dev/null | grep '"tc0"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v mix:', d['med_us'], d['min_us'], d['batch_us'])"
done
cp /tmp/libtapir_saved.so tapnet_amd/csrc/libtapir_hip.so
