"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/tapnet_head.npz by EXECUTING the reference's own
``tapnet/models/tapnet_model.py`` (TAPNet.__call__ :173-290, tracks_from_cost_volume :111-171) and
``tapnet/utils/model_utils.py`` (interp :176-206, soft_argmax_heatmap :209-247, heatmaps_to_points :250-314)
from /root/reference over the numpy stand-ins of ``oracle/hk_numpy_shim.py`` (JAX / Haiku are not installable
in the build container; see that file for what is the reference's and what is restated).

  python oracle/make_tapnet_golden.py            # writes tests/golden/tapnet_head.npz
  python oracle/make_tapnet_golden.py --check    # re-runs the reference and compares with the committed file

Cases (feature grids are L2-normalised random [B,T,h,w,256]; `feature_grid=` is passed, so the TSM-ResNet
backbone -- out of scope, SURVEY.md 8 -- never runs):
  a: num_heads 1, 1 clip x 3 frames, 128x128 video (16x16 grid), 12 queries in ragged chunks of 5
  b: num_heads 1, 2 clips x 2 frames, 96x128 video (12x16 grid), 6 queries, get_query_feats
  c: num_heads 2, 1 clip x 2 frames, 64x64 video (8x8 grid), 5 queries          (restatement only)
Parameters are created by the stand-in's hk.transform_with_state(...).init in HAIKU layout and under the names
the reference's module tree gives them ('tap_net/~/<module>': {'w','b'}; Conv3D kernels [1,3,3,in,out], Linear
[in,out]) and stored under those names: the tests convert them with the product's
tapnet_amd.tapnet_model.from_haiku_params, so the converter is inside the pin.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REFERENCE_ROOT = os.environ.get('TAPNET_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'tapnet_head.npz')


def make_peaky(p):
  """Peaky heat maps (a trained head's are): channel 0 of hid1 passes the cost volume through, hid2 reads it."""
  w1, w2 = p['tap_net/~/cost_volume_regression_1'], p['tap_net/~/cost_volume_regression_2']
  w1['w'][..., 0] = 0.0
  w1['w'][0, 1, 1, :, 0] = 1.0
  w1['b'][0] = 0.0
  w2['w'] *= np.float32(0.1)
  w2['w'][0, :, :, 0, 0] = 0.0
  w2['w'][0, 1, 1, 0, 0] = 3.0


def l2n(x):
  return (x / np.sqrt(np.maximum(np.sum(np.square(x), -1, keepdims=True), 1e-12))).astype(np.float32)


CASES = dict(a=dict(heads=1, B=1, T=3, hw=(16, 16), Q=12, chunk=5, seed=1),
             b=dict(heads=1, B=2, T=2, hw=(12, 16), Q=6, chunk=6, seed=2),
             c=dict(heads=2, B=1, T=2, hw=(8, 8), Q=5, chunk=2, seed=3))


def run_reference():
  from oracle import hk_numpy_shim as shim
  shim.install()
  if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)
  import haiku as hk                                  # the stand-in
  from tapnet.models import tapnet_model as mod       # the reference, imported over the stand-ins
  out = {}
  for tag, c in CASES.items():
    rng = np.random.default_rng(c['seed'])
    h, w = c['hw']
    H, W = 8 * h, 8 * w
    grid = l2n(rng.standard_normal((c['B'], c['T'], h, w, 256)))
    qp = np.stack([rng.integers(0, c['T'], (c['B'], c['Q'])), rng.uniform(0, H, (c['B'], c['Q'])),
                   rng.uniform(0, W, (c['B'], c['Q']))], -1).astype(np.float32)
    video = np.zeros((c['B'], c['T'], H, W, 3), np.float32)

    def forward(video, qp, grid):
      model = mod.TAPNet(num_heads=c['heads'])
      return model(video, False, qp, query_chunk_size=c['chunk'], get_query_feats=True, feature_grid=grid)

    fn = hk.transform_with_state(forward)
    params, state = fn.init(np.array([0, c['seed']]), video, qp, grid)   # names as the module tree creates them
    make_peaky(params)
    res, _ = fn.apply(params, state, None, video, qp, grid)
    for k, v in params.items():
      for kk, a in v.items():
        out[f'{tag}/params/{k}/{kk}'] = a
    out[f'{tag}/feature_grid'] = grid
    out[f'{tag}/query_points'] = qp
    out[f'{tag}/video_shape'] = np.array(video.shape)
    out[f'{tag}/num_heads'] = np.array(c['heads'])
    for k in ('tracks', 'occlusion', 'query_feats'):
      out[f'{tag}/{k}'] = np.asarray(res[k], np.float32)
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--check', action='store_true')
  a = ap.parse_args()
  out = run_reference()
  if a.check:
    ref = np.load(OUT)
    assert sorted(ref.files) == sorted(out), 'key sets differ'
    worst = max(float(np.max(np.abs(ref[k].astype(np.float64) - out[k]))) for k in out)
    print('max |committed - regenerated| =', worst)
    assert worst < 1e-6
    return
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, os.path.getsize(OUT), 'bytes')


if __name__ == '__main__':
  main()
