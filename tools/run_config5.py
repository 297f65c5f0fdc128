#!/usr/bin/env python
"""BASELINE.json configs[4] on ONE GPU: 512x512x96 clip, 4096 queries, default refinement
resolutions [(256,256),(512,512)] -> 8 refinement iterations.  Checks that the
path scales (393216 token rows, 1.5 GB of feature grids), that results are finite, that a query
subset reproduces, and reports the time."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tapnet_amd import synthetic, tapir_model

T, S, Q = 96, 512, 4096
w = synthetic.make_weights(0, 1, True)     # BootsTAPIR kwargs
m = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, softmax_temperature=10.0, weights=w,
                      dtype='bfloat16', device='cuda:0')
video = torch.as_tensor(synthetic.make_video(3, T, S, S)).cuda()
qp = torch.as_tensor(synthetic.make_queries(4, Q, T, S, S)).cuda()
out = m(video, False, qp)
torch.cuda.synchronize()
assert torch.isfinite(out['tracks']).all() and out['tracks'].shape == (1, Q, T, 2)
assert len(out['unrefined_tracks']) == 8
t0 = time.perf_counter()
fg = m.get_feature_grids(video)
torch.cuda.synchronize()
t_bb = time.perf_counter() - t0
t0 = time.perf_counter()
out = m(video, False, qp, feature_grids=fg)
torch.cuda.synchronize()
t_hot = time.perf_counter() - t0
m.profile_enable(True); m.profile_read()
m(video, False, qp, feature_grids=fg)
torch.cuda.synchronize()
prof = {k: (round(v[0], 3), v[1]) for k, v in m.profile_read().items() if v[1]}
m.profile_enable(False)
sub = m(video, False, qp[:, 1000:1256], feature_grids=fg)
err = float((sub['tracks'] - out['tracks'][:, 1000:1256]).abs().max())
print(json.dumps(dict(config='512x512x96, Q=4096, BootsTAPIR kwargs, 8 iterations, bf16, 1 GPU',
                      backbone_s=round(t_bb, 4), hot_path_s=round(t_hot, 4),
                      points_per_s=round(Q / (t_bb + t_hot), 1),
                      point_frames_per_s=round(Q * T / (t_bb + t_hot), 1),
                      subset_max_abs_diff_px=err, kernels_ms_launches=prof,
                      peak_mem_GB=round(torch.cuda.max_memory_allocated() / 2**30, 2))))
