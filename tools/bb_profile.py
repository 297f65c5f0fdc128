#!/usr/bin/env python
"""Steady-state backbone only, for rocprofv3 --kernel-trace --stats: warms up (MIOpen solver search), then
runs get_feature_grids `reps` times between two roctx-free markers (kernel names with >= reps calls are
the steady-state ones)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tapnet_amd import synthetic, tapir_model
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
w = synthetic.make_weights(0, 0, False)
m = tapir_model.TAPIR(pyramid_level=0, extra_convs=False, weights=w, dtype='bfloat16', device='cuda:0')
video = torch.as_tensor(synthetic.make_video(1, 48, 256, 256)).cuda()
for _ in range(3):
  m.get_feature_grids(video)
torch.cuda.synchronize()
for _ in range(reps):
  m.get_feature_grids(video)
torch.cuda.synchronize()
