"""Host logic of tapnet_amd.bulk_tracking (the track_many_points workload of
tapnet/robotap/tapir_clustering.py:918-1179) with a stub model / tracker: sampling order, batching,
padding of the last batch, per-episode slicing.  The GPU run is in test_gpu_parity.py."""
import numpy as np
import pytest

torch = pytest.importorskip('torch')

from tapnet_amd import bulk_tracking as bt
from tapnet_amd.tapir_model import QueryFeatures


class StubModel:
  use_causal_conv = True
  device = torch.device('cpu')

  def _dev(self, x, dtype=torch.float32):
    return torch.as_tensor(x).to(dtype)

  def get_feature_grids(self, frames):
    return frames   # [1,F,H,W,3]

  def get_query_features(self, frames, is_training, qp, fg):
    # "features" = (mean of the query's frame, y, x): lets the test see which frame each point used
    f = qp[0, :, 0].long()
    mean = fg[0].mean(dim=(1, 2, 3))[f]
    lo = torch.stack([mean, qp[0, :, 1], qp[0, :, 2]], dim=-1)[None]
    return QueryFeatures((lo,), (lo * 2,), ((frames.shape[2], frames.shape[3]),))


class StubTracker:
  """tracks = the query's (x, y) + frame mean; occlusion logit -5 (visible) for even points"""
  instances = 0

  def __init__(self, model, n, hw):
    self.model, self.n, self.hw = model, n, hw
    StubTracker.instances += 1

  def set_query_features(self, qf):
    assert qf.lowres[0].shape[1] == self.n
    self.qf = qf

  def step(self, frame):
    lo = self.qf.lowres[0][0]
    m = frame.mean()
    tracks = torch.stack([lo[:, 2] + m, lo[:, 1] + m], dim=-1)[None, :, None]
    occ = torch.where(torch.arange(self.n) % 2 == 0, -5.0, 5.0)[None, :, None]
    return dict(tracks=tracks, occlusion=occ, expected_dist=torch.full_like(occ, -5.0))


def _videos():
  rng = np.random.default_rng(0)
  return {'a': rng.integers(0, 255, (9, 16, 20, 3), dtype=np.uint8),
          'b': rng.integers(0, 255, (6, 16, 20, 3), dtype=np.uint8)}


def test_sampling_follows_the_reference_draw_order():
  s = bt.sample_query_points([(9, 16, 20, 3), (6, 16, 20, 3)], 4, 3)
  assert [(v, i) for v, i, _ in s] == [(0, 0), (0, 4), (0, 8), (1, 0), (1, 4)]
  np.random.seed(42)   # the reference's generator (tapir_clustering.py:1059)
  for _, _, yx in s:
    qp = np.random.uniform(0.0, 1.0, [3, 3]) * np.array([0.0, 16 * 0.8, 20 * 0.8])[None] + \
        np.array([0.0, 16 * 0.1, 20 * 0.1])[None]
    np.testing.assert_array_equal(yx, qp[:, 1:])
  assert all((yx[:, 0] >= 1.6).all() and (yx[:, 0] <= 14.4).all() for _, _, yx in s)


def test_join_count_slice_visibility():
  a = QueryFeatures((np.zeros((1, 3, 4)),), (np.zeros((1, 3, 2)),), ((8, 8),))
  b = QueryFeatures((np.ones((1, 5, 4)),), (np.ones((1, 5, 2)),), ((8, 8),))
  j = bt.query_features_join([a, b])
  assert bt.query_features_count(j) == 8 and j.hires[0].shape == (1, 8, 2)
  assert bt.query_features_slice(j, 3, 8).lowres[0].min() == 1
  pred = dict(tracks=np.zeros((1, 2, 1, 2)), occlusion=np.array([[[0.0], [100.0]]]),
              expected_dist=np.array([[[-100.0], [0.0]]]))
  trk, vis = bt.predictions_to_tracks_visibility(pred)
  assert trk.shape == (2, 2)
  np.testing.assert_allclose(vis, [0.5, 0.0], atol=1e-6)
  np.testing.assert_allclose(bt.preprocess_frames(np.array([0, 255], np.uint8)), [-1.0, 1.0])


def test_track_many_points_bookkeeping():
  videos = _videos()
  StubTracker.instances = 0
  # 5 sampled frames x 2 points = 10 points, batches of 4 -> 3 batches, the last padded by 2
  out = bt.track_many_points(videos, ['a', 'b'], StubModel(), frame_stride=4, points_per_frame=2,
                             point_batch_size=4, tracker_factory=StubTracker)
  assert StubTracker.instances == 1
  assert out['separation_tracks']['a'].shape == (10, 9, 2) and out['separation_tracks']['b'].shape == (10, 6, 2)
  assert out['separation_visibility']['b'].shape == (10, 6) and out['separation_visibility']['a'].dtype == bool
  assert out['video_shape'] == {'a': (9, 16, 20, 3), 'b': (6, 16, 20, 3)}
  v, f, yx = out['query_points']
  np.testing.assert_array_equal(v, [0] * 6 + [1] * 4)
  np.testing.assert_array_equal(f, [0, 0, 4, 4, 8, 8, 0, 0, 4, 4])
  # every point's "feature" holds the mean of the frame it was sampled from, and its own (y, x)
  lo = out['query_features'].lowres[0][0]
  for k in range(10):
    frame = bt.preprocess_frames(videos['ab'[v[k]]][f[k]])
    assert abs(lo[k, 0] - frame.mean()) < 1e-5
    np.testing.assert_allclose(lo[k, 1:], yx[k], rtol=1e-6)
  # tracks: (x, y) + mean of the tracked frame, for every point in every episode
  for name in 'ab':
    for t in range(videos[name].shape[0]):
      m = bt.preprocess_frames(videos[name][t]).mean()
      np.testing.assert_allclose(out['separation_tracks'][name][:, t], yx[:, ::-1] + m, rtol=1e-5, atol=1e-5)
  # visibility pattern of the stub (even slots of each batch visible): batches [0..3] [4..7] [8, 9 | pad]
  np.testing.assert_array_equal(out['separation_visibility']['a'][:, 0], [True, False] * 5)


def test_requires_causal_model_and_whole_frames():
  m = StubModel(); m.use_causal_conv = False
  with pytest.raises(ValueError):
    bt.track_many_points(_videos(), ['a'], m, tracker_factory=StubTracker)
  with pytest.raises(ValueError):
    bt.track_many_points(_videos(), ['a'], StubModel(), points_per_frame=3, point_batch_size=4,
                         tracker_factory=StubTracker)
