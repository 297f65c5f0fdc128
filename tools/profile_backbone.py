"""Runs Backbone.features eagerly (no hipGraph) N times: meant to be run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split of the backbone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tapnet_amd import synthetic
from tapnet_amd.tapir_model import TAPIR

w = synthetic.make_weights(0, 0, False)
m = TAPIR(pyramid_level=0, extra_convs=False, softmax_temperature=20.0, weights=w, device='cuda:0', dtype='bfloat16')
bb = m._backbone
bb.graph_min_frames = 0
frames = torch.rand(48, 256, 256, 3, device='cuda:0') * 2 - 1
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
  bb.features(frames)
torch.cuda.synchronize()
