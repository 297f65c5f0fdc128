#!/usr/bin/env python
"""Where does a full TAPIR.__call__ step spend its time: GPU events around the stages + host time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tapnet_amd import synthetic, tapir_model

w = synthetic.make_weights(0, 0, False)
T, S, Q = 48, 256, 256
video = torch.as_tensor(synthetic.make_video(1, T, S, S)).cuda()
qp = torch.as_tensor(synthetic.make_queries(2, Q, T, S, S)).cuda()
m = tapir_model.TAPIR(pyramid_level=0, weights=w, dtype='bfloat16', device='cuda:0')
for _ in range(3):
  m(video, False, qp)
torch.cuda.synchronize()

def ev():
  e = torch.cuda.Event(enable_timing=True); e.record(); return e

N = 10
acc = {}
host = {}
t_all0 = time.perf_counter()
for _ in range(N):
  h0 = time.perf_counter(); e0 = ev()
  fg = m.get_feature_grids(video)
  h1 = time.perf_counter(); e1 = ev()
  qf = m.get_query_features(video, False, qp, fg)
  h2 = time.perf_counter(); e2 = ev()
  traj = m.estimate_trajectories((S, S), False, fg, qf, qp)
  h3 = time.perf_counter(); e3 = ev()
  out = torch.mean(torch.stack(traj['tracks'][4::4]), dim=0)
  h4 = time.perf_counter(); e4 = ev()
  torch.cuda.synchronize()
  for k, a, b in (('backbone', e0, e1), ('query_feats', e1, e2), ('estimate', e2, e3), ('post', e3, e4)):
    acc[k] = acc.get(k, 0) + a.elapsed_time(b)
  for k, a, b in (('backbone', h0, h1), ('query_feats', h1, h2), ('estimate', h2, h3), ('post', h3, h4)):
    host[k] = host.get(k, 0) + (b - a) * 1e3
print('per step, GPU ms :', {k: round(v / N, 3) for k, v in acc.items()}, 'sum', round(sum(acc.values()) / N, 3))
print('per step, host ms:', {k: round(v / N, 3) for k, v in host.items()}, 'sum', round(sum(host.values()) / N, 3))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
  m(video, False, qp)
torch.cuda.synchronize()
print('full __call__ back to back:', round((time.perf_counter() - t0) / N * 1e3, 3), 'ms')
