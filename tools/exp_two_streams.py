#!/usr/bin/env python
"""Experiment: does running two half-size query groups on two HIP streams (MFMA-bound GEMMs of one
group against the HBM/VALU-bound kernels of the other) beat one full-size pass?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tapnet_amd import synthetic, tapir_model

w = synthetic.make_weights(0, 0, False)
T, S, Q = 48, 256, 256
video = torch.as_tensor(synthetic.make_video(1, T, S, S)).cuda()
qp = torch.as_tensor(synthetic.make_queries(2, Q, T, S, S)).cuda()
models = [tapir_model.TAPIR(pyramid_level=0, weights=w, dtype='bfloat16', device='cuda:0') for _ in range(4)]
fg = models[0].get_feature_grids(video)
streams = [torch.cuda.Stream() for _ in range(4)]

def run(groups):
  n = Q // groups
  ev = torch.cuda.Event(); ev.record()
  for g in range(groups):
    with torch.cuda.stream(streams[g]):
      streams[g].wait_event(ev)
      models[g](video, False, qp[:, g * n:(g + 1) * n], feature_grids=fg)
  for g in range(groups):
    torch.cuda.current_stream().wait_stream(streams[g])

for groups in (1, 2, 4, 1, 2, 4):
  for _ in range(3):
    run(groups)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(10):
    run(groups)
  torch.cuda.synchronize()
  print(f'groups={groups}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per hot path (Q={Q})', flush=True)
