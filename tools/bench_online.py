#!/usr/bin/env python
"""Per-frame latency of the online (causal) tracker, BASELINE.json configs[3]: 256x256 frames,
256 query points, 4 refinement iterations per frame; eager launches vs the hipGraph replay.

    python tools/bench_online.py [--queries 256] [--frames 60] [--dtype bf16]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from tapnet_amd import online, synthetic, tapir_model


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--queries', type=int, default=256)
  ap.add_argument('--frames', type=int, default=60)
  ap.add_argument('--size', type=int, default=256)
  ap.add_argument('--dtype', default='bf16')
  ap.add_argument('--eager-only', action='store_true', help='one configuration, eager launches (kernel traces)')
  ap.add_argument('--gemm-mode', type=int, default=3, help='0 = split-K + reduce pair (round 2), 1 = one-launch few-row GEMMs, 2 = 1 + the channel MLP of a block in one launch, 3 = 2 + the whole mixer of a frame as one persistent launch (the engine default)')
  args = ap.parse_args()
  dtype = 'bfloat16' if args.dtype == 'bf16' else 'float32'
  w = synthetic.make_weights(0, pyramid_level=1, extra_convs=True)   # the causal checkpoint's kwargs
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, use_causal_conv=True, weights=w,
                        dtype=dtype, device='cuda:0')
  assert m._lib.tapir_debug_set_gemm_mode(m._ctx, args.gemm_mode) == 0
  S, Q = args.size, args.queries
  video = torch.as_tensor(synthetic.make_video(1, 8, S, S)).cuda()
  qp = torch.as_tensor(synthetic.make_queries(2, Q, 1, S, S)).cuda()
  rows = []
  checked = {}
  configs = (('auto', False),) if args.eager_only else (('auto', False), ('auto', True), ('miopen', True), ('hip', True))
  for conv_mode, use_graph in configs:
    if m._backbone.dtype != torch.bfloat16 and conv_mode == 'hip':
      continue
    m._backbone.conv_mode = conv_mode
    trk = online.OnlineTracker(m, Q, (S, S), use_graph=use_graph)
    trk.init(video[:, :1], qp)
    for t in range(5):
      trk.step(video[:, t % 8:t % 8 + 1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(args.frames):
      out = trk.step(video[:, t % 8:t % 8 + 1])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.frames * 1e3
    assert torch.isfinite(out['tracks']).all()
    trk.check()
    # every frame of a fresh session finite and equal across the configurations of one backbone mode (a replay that skipped work
    # would be fast and wrong: profiles/r06_graph_memset_hazard.txt)
    trk.init(video[:, :1], qp)
    frames = [trk.step(video[:, t % 8:t % 8 + 1])['tracks'].clone() for t in range(12)]
    trk.check()
    assert all(bool(torch.isfinite(f).all()) for f in frames)
    ref = checked.setdefault(conv_mode, frames)
    worst = max(float((a - b).abs().max()) for a, b in zip(frames, ref))
    assert worst < 1e-3, worst
    rows.append(dict(mode='hipGraph replay' if use_graph else 'eager launches', backbone_convs=conv_mode, gemm_mode=args.gemm_mode, ms_per_frame=round(ms, 3),
                     frames_per_s=round(1e3 / ms, 1), points_frames_per_s=round(Q * 1e3 / ms, 1)))
    print(json.dumps(dict(workload=f'online TAPIR {S}x{S}, Q={Q}, 4 iters/frame, {dtype}', **rows[-1])), flush=True)


if __name__ == '__main__':
  main()
