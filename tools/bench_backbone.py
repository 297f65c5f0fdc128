"""Times Backbone.features (R7) for one clip: the frames on one stream, or cut into groups of frames on
several HIP streams (tapnet_amd/backbone.py: `streams`).  usage: python tools/bench_backbone.py [--frames 48]"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tapnet_amd import synthetic
from tapnet_amd.tapir_model import TAPIR


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--frames', type=int, default=48)
  ap.add_argument('--size', type=int, default=256)
  ap.add_argument('--dtype', default='bfloat16')
  ap.add_argument('--reps', type=int, default=20)
  a = ap.parse_args()
  w = synthetic.make_weights(0, 0, False)
  m = TAPIR(pyramid_level=0, extra_convs=False, softmax_temperature=20.0, weights=w, device='cuda:0', dtype=a.dtype)
  bb = m._backbone
  frames = torch.rand(a.frames, a.size, a.size, 3, device='cuda:0') * 2 - 1
  ap_sets = [('all HIP', {'stem', 'conv_0', 'conv_1', 'conv_0_s2', 'proj_conv', 'proj_conv_s2'}),
             ('all block convolutions, MIOpen stem', {'conv_0', 'conv_1', 'conv_0_s2', 'proj_conv', 'proj_conv_s2'}),
             ('3x3 stride 1 only', {'conv_0', 'conv_1'}),
             ('+ stride-2 3x3', {'conv_0', 'conv_1', 'conv_0_s2'}),
             ('+ 1x1 stride 1', {'conv_0', 'conv_1', 'proj_conv'}),
             ('+ 1x1 stride 2', {'conv_0', 'conv_1', 'proj_conv_s2'}),
             ('none (MIOpen)', set())]
  for name, kinds in ap_sets:
    bb.hip_convs = kinds
    bb.streams = 1
    for _ in range(4):
      bb.features(frames)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
      bb.features(frames)
    e1.record()
    torch.cuda.synchronize()
    print(json.dumps({'workload': f'Backbone.features {a.frames}x{a.size}x{a.size} {a.dtype}', 'hip_convs': name,
                      'ms': round(e0.elapsed_time(e1) / a.reps, 3)}), flush=True)
  bb.hip_convs = ap_sets[0][1]
  ref = None
  for streams, chunk in ((1, None), (2, None), (4, None), (1, 12), (2, 12)):
    bb.streams = streams
    for _ in range(4):
      out = bb.features(frames, chunk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
      out = bb.features(frames, chunk)
    e1.record()
    torch.cuda.synchronize()
    if ref is None:
      ref = out
    same = all(torch.equal(x, y) for x, y in zip(out, ref))
    print(json.dumps({'workload': f'Backbone.features {a.frames}x{a.size}x{a.size} {a.dtype}', 'streams': streams, 'frames_per_group': chunk,
                      'ms': round(e0.elapsed_time(e1) / a.reps, 3), 'bit_identical_to_1_stream': same}), flush=True)


if __name__ == '__main__':
  main()
