#!/usr/bin/env python
"""One steady-state TAPIR.__call__ (config 2, bf16) for rocprofv3 --kernel-trace: warms up, then runs
`reps` calls; tools/timeline.py prints the last call's kernel timeline, the sum of kernel durations and
the idle gaps between kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tapnet_amd import synthetic, tapir_model
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
w = synthetic.make_weights(0, 0, False)
m = tapir_model.TAPIR(pyramid_level=0, extra_convs=False, weights=w, dtype='bfloat16', device='cuda:0')
video = torch.as_tensor(synthetic.make_video(1, 48, 256, 256)).cuda()
qp = torch.as_tensor(synthetic.make_queries(101, 256, 48, 256, 256)).cuda()
for _ in range(3):
  m(video, False, qp)
torch.cuda.synchronize()
for _ in range(reps):
  m(video, False, qp)
  torch.cuda.synchronize()
