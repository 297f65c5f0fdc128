"""Re-runs given cases of tools/fuzz_parity.py (--seed S, --tmax, --qmax, case indices) and attributes deviations: per
(query, frame) the deviation of the cost-volume initialisation and of the final tracks, next to the oracle's relative
gap between the two largest soft-max cells (a near-tie arg max may legitimately flip between two f32 implementations)."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import backbone_torch, tapir_oracle as O
from tapnet_amd import synthetic, tapir_model

ap = argparse.ArgumentParser()
ap.add_argument('--seed', type=int, default=7); ap.add_argument('--tmax', type=int, default=50)
ap.add_argument('--qmax', type=int, default=200); ap.add_argument('--cases', type=int, nargs='+', default=[1, 6])
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
for i in range(max(a.cases) + 1):
  pyr, extra = int(rng.integers(0, 2)), bool(rng.integers(0, 2))
  T, Q = int(rng.integers(1, a.tmax + 1)), int(rng.integers(1, a.qmax + 1))
  ih, iw = 8 * int(rng.integers(6, 13)), 8 * int(rng.integers(6, 13))
  scale = float(rng.choice([1.0, 1.0, 1.5, 2.0]))
  H, W = 8 * int(round(ih * scale / 8)), 8 * int(round(iw * scale / 8))
  chunk = int(rng.choice([Q, max(1, Q // 2), 7]))
  seed = int(rng.integers(0, 1 << 30))
  if i not in a.cases:
    continue
  w = synthetic.make_weights(seed % 1000, pyr, extra)
  video = synthetic.make_video(seed, T, H, W).astype(np.float32)
  qp = synthetic.make_queries(seed + 1, Q, T, H, W).astype(np.float32)
  m = tapir_model.TAPIR(pyramid_level=pyr, extra_convs=extra, initial_resolution=(ih, iw), weights=w, device='cuda:0')
  fg = m.get_feature_grids(video, False)
  lows = [x.cpu().numpy() for x in fg.lowres]; his = [x.cpu().numpy() for x in fg.hires]
  res = [tuple(r) for r in fg.resolutions]
  out = m(video, False, qp, query_chunk_size=chunk, feature_grids=fg)
  # oracle on the ENGINE's grids: isolates the hot path
  ref = O.tapir_from_grids(w, video.shape, lows, his, res, qp, pyramid_level=pyr, softmax_temperature=20.0,
                           initial_resolution=(ih, iw))
  ql, _ = O.get_query_features(lows, his, res, qp, video.shape)
  _, _, _, st = O.tracks_from_cost_volume(w, ql[0], lows[0], None, (ih, iw), 20.0, return_stages=True)
  gap = st['top2_rel_gap']
  d0 = np.linalg.norm(np.asarray(out['unrefined_tracks'][0]) - ref['unrefined_tracks'][0], axis=-1)
  dF = np.linalg.norm(np.asarray(out['tracks']) - ref['tracks'], axis=-1)
  bad = np.argwhere(dF > 5e-3)
  print(json.dumps(dict(case=i, T=T, Q=Q, init=(ih, iw), video=(H, W), pyr=pyr, extra=extra, hot_path_only=True,
                        init_dev_max=float(d0.max()), final_dev_max=float(dF.max()), n_bad=int(len(bad)),
                        bad=[dict(q=int(q), t=int(t), init_dev=float(d0[b, q, t]), final_dev=float(dF[b, q, t]),
                                  top2_rel_gap=float(gap[b, q, t])) for b, q, t in bad[:8]],
                        gap_min=float(gap.min()))), flush=True)
