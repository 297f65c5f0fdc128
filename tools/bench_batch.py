#!/usr/bin/env python
"""Throughput of TAPIR.__call__ on a BATCH of clips (video [B,48,256,256,3], B x 256 queries): with B >= 2 the
mixer sees > 256 tracks and the engine picks the wide kernel (two tracks per workgroup share every weight fragment),
the backbone's launches cover B x 48 frames.  Secondary number next to bench.py's B = 1 headline."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tapnet_amd import synthetic, tapir_model

w = synthetic.make_weights(0, 0, False)
m = tapir_model.TAPIR(pyramid_level=0, extra_convs=False, softmax_temperature=20.0, weights=w, dtype='bfloat16', device='cuda:0')
T, S, Q = 48, 256, 256
for B in (1, 2, 4):
  video = torch.as_tensor(synthetic.make_video(1, T, S, S, batch=B)).cuda()
  qp = torch.as_tensor(synthetic.make_queries(2, Q, T, S, S, batch=B)).cuda()
  for _ in range(5):
    out = m(video, False, qp)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  n = 20
  for _ in range(n):
    out = m(video, False, qp)
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / n
  assert torch.isfinite(out['tracks']).all()
  print(json.dumps(dict(workload=f'TAPIR bf16, batch of {B} clips 256x256x48, Q=256 each', ms_per_call=round(dt * 1e3, 3),
                        ms_per_clip=round(dt * 1e3 / B, 3), points_per_s=round(B * Q / dt, 1))), flush=True)
