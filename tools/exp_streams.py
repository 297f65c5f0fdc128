"""Backbone.features for a 48-frame clip with 1..8 frame groups on as many streams (hipGraph replay on)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tapnet_amd import synthetic
from tapnet_amd.tapir_model import TAPIR
for extra in (False, True):
  w = synthetic.make_weights(0, 1 if extra else 0, extra)
  m = TAPIR(pyramid_level=1 if extra else 0, extra_convs=extra, weights=w, device='cuda:0', dtype='bfloat16')
  bb = m._backbone
  frames = torch.rand(48, 256, 256, 3, device='cuda:0') * 2 - 1
  for streams in (2, 4, 6, 3, 4, 2, 8):
    bb.streams = streams
    for _ in range(5):
      bb.features(frames, borrow=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
      bb.features(frames, borrow=True)
    e1.record(); torch.cuda.synchronize()
    print(json.dumps(dict(extra_convs=extra, streams=streams, ms=round(e0.elapsed_time(e1) / 30, 3))), flush=True)
