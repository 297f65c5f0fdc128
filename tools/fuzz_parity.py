"""Random-shape parity sweep on the GPU box: the f32 engine, video -> tracks, against the oracle (plain-PyTorch backbone +
numpy hot path) on small random configurations -- frame counts, query counts, non-square frame sizes, multi-resolution,
TAPIR / BootsTAPIR kwargs, query chunking.  A tool (tests/ holds the fixed cases); prints one line per case and the worst.

    python tools/fuzz_parity.py --cases 30 --seed 0
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import backbone_torch, tapir_oracle as O   # noqa: E402  (checker only)
from tapnet_amd import synthetic, tapir_model           # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--cases', type=int, default=20)
  ap.add_argument('--seed', type=int, default=0)
  ap.add_argument('--tmax', type=int, default=20)
  ap.add_argument('--qmax', type=int, default=24)
  a = ap.parse_args()
  rng = np.random.default_rng(a.seed)
  worst = 0.0
  for i in range(a.cases):
    pyr, extra = int(rng.integers(0, 2)), bool(rng.integers(0, 2))
    T, Q = int(rng.integers(1, a.tmax + 1)), int(rng.integers(1, a.qmax + 1))
    ih, iw = 8 * int(rng.integers(6, 13)), 8 * int(rng.integers(6, 13))          # initial_resolution 48..96
    scale = float(rng.choice([1.0, 1.0, 1.5, 2.0]))
    H, W = 8 * int(round(ih * scale / 8)), 8 * int(round(iw * scale / 8))
    chunk = int(rng.choice([Q, max(1, Q // 2), 7]))
    seed = int(rng.integers(0, 1 << 30))
    w = synthetic.make_weights(seed % 1000, pyr, extra)
    video = synthetic.make_video(seed, T, H, W).astype(np.float32)
    qp = synthetic.make_queries(seed + 1, Q, T, H, W).astype(np.float32)
    t0 = time.time()
    m = tapir_model.TAPIR(pyramid_level=pyr, extra_convs=extra, initial_resolution=(ih, iw), weights=w, device='cuda:0')
    out = m(video, False, qp, query_chunk_size=chunk)
    res = [(ih, iw)] + [tuple(r) for r in O.generate_default_resolutions((H, W), (ih, iw))]
    bb = backbone_torch.TorchBackbone(w, extra)
    lows, his, cur, lo, hi = [], [], None, None, None
    for r in res:
      if r != cur:
        v = torch.as_tensor(video)
        if r != (H, W):   # the torch twin's resize (no antialias): what the engine does for torch-named weights
          v = torch.nn.functional.interpolate(v[0].permute(0, 3, 1, 2), size=r, mode='bilinear', align_corners=False
                                              ).permute(0, 2, 3, 1)[None]
        l, h = bb.features(v.reshape(-1, r[0], r[1], 3))
        lo, hi, cur = l.numpy()[None], h.numpy()[None], r
      lows.append(lo); his.append(hi)
    ref = O.tapir_from_grids(w, video.shape, lows, his, res, qp, pyramid_level=pyr, softmax_temperature=20.0,
                             initial_resolution=(ih, iw))
    d = {k: float(np.abs(np.asarray(out[k]) - ref[k]).max()) for k in ('tracks', 'occlusion', 'expected_dist')}
    worst = max(worst, d['tracks'])
    print(json.dumps(dict(case=i, pyr=pyr, extra=extra, T=T, Q=Q, init=(ih, iw), video=(H, W), levels=len(res), chunk=chunk,
                          **{k: round(v, 6) for k, v in d.items()}, s=round(time.time() - t0, 1))), flush=True)
    del m
  print('worst tracks deviation', worst)


if __name__ == '__main__':
  main()
