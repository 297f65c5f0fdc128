"""tapnet_amd.tapvid (query sampling + TAP-Vid metrics) against outputs of the reference's own
functions on seeded tracks (tests/golden/tapvid_metrics.npz, oracle/make_tapvid_golden.py)."""
import os

import numpy as np
import pytest

from tapnet_amd import tapvid
from tests.golden_util import GOLDEN_DIR


@pytest.fixture(scope='module')
def golden():
  return np.load(os.path.join(GOLDEN_DIR, 'tapvid_metrics.npz'))


@pytest.mark.parametrize('mode', ['strided', 'first'])
def test_query_sampling(golden, mode):
  frames = np.zeros((golden['occ'].shape[1], 2, 2, 3), np.float32)
  sampler = tapvid.sample_queries_strided if mode == 'strided' else tapvid.sample_queries_first
  ex = sampler(golden['occ'], golden['pts'], frames)
  for k in ('query_points', 'target_points', 'occluded'):
    np.testing.assert_array_equal(ex[k], golden[f'{mode}_{k}'])


@pytest.mark.parametrize('mode', ['strided', 'first'])
@pytest.mark.parametrize('trackwise', [False, True])
def test_metrics(golden, mode, trackwise):
  with np.errstate(divide='ignore', invalid='ignore'):
    m = tapvid.compute_tapvid_metrics(golden[f'{mode}_query_points'], golden[f'{mode}_occluded'],
                                      golden[f'{mode}_target_points'], golden[f'{mode}_pred_occ'],
                                      golden[f'{mode}_pred_tracks'], mode, get_trackwise_metrics=trackwise)
  pre = f'{mode}_{"tw_" if trackwise else ""}'
  keys = ['occlusion_accuracy', 'average_jaccard', 'average_pts_within_thresh'] + \
         [f'{a}_{t}' for a in ('pts_within', 'jaccard') for t in tapvid.THRESHOLDS]
  for k in keys:
    np.testing.assert_allclose(m[k], golden[pre + k], rtol=1e-12, atol=0, equal_nan=True, err_msg=k)


def test_unknown_mode(golden):
  with pytest.raises(ValueError):
    tapvid.compute_tapvid_metrics(golden['first_query_points'], golden['first_occluded'],
                                  golden['first_target_points'], golden['first_pred_occ'],
                                  golden['first_pred_tracks'], 'nope')


def test_postprocess_occlusions():
  occ = np.array([-5.0, 5.0, -5.0, 0.0])
  expd = np.array([-5.0, -5.0, 5.0, 0.0])
  np.testing.assert_array_equal(tapvid.postprocess_occlusions(occ, expd), [False, True, True, True])


def test_evaluate_driver_with_a_perfect_tracker():
  """tapvid.evaluate (evaluation_datasets.py:48-192 metrics + the strided / first query samplers) end
  to end on synthetic ground truth: a stand-in model that returns the ground truth scores AJ = 1,
  one that predicts everything occluded scores AJ = 0."""
  import numpy as np
  from tapnet_amd import tapvid
  rng = np.random.default_rng(0)
  T, N, H = 12, 7, 64
  pts = rng.uniform(4, H - 4, (N, T, 2))
  occ = rng.random((N, T)) < 0.2
  occ[:, 0] = False
  video = rng.uniform(-1, 1, (T, H, H, 3)).astype(np.float32)

  class Oracle:   # answers every query with its ground-truth track
    def __init__(self, ex, blind=False): self.ex, self.blind = ex, blind
    def get_feature_grids(self, video): return None
    def __call__(self, video, is_training, qp, feature_grids=None):
      ex = self.ex
      idx = [int(np.argmin(np.abs(ex['query_points'][0] - q).sum(-1))) for q in qp[0]]
      occl = np.where(ex['occluded'][0][idx] | self.blind, 20.0, -20.0)
      return dict(tracks=ex['target_points'][:, idx], occlusion=occl[None], expected_dist=np.full_like(occl, -20.0)[None])

  for mode, sampler in (('strided', tapvid.sample_queries_strided), ('first', tapvid.sample_queries_first)):
    ex = sampler(occ, pts, video)
    good = tapvid.evaluate(Oracle(ex), [('clip', ex)], query_mode=mode, query_chunk=5)
    assert abs(good['average_jaccard'] - 1.0) < 1e-9 and abs(good['occlusion_accuracy'] - 1.0) < 1e-9
    bad = tapvid.evaluate(Oracle(ex, blind=True), [('clip', ex)], query_mode=mode)
    assert bad['average_jaccard'] == 0.0


def test_davis_examples_resize_is_uint8_lanczos(tmp_path):
  """tapvid/evaluation_datasets.py:41-45: frames are resized as uint8 with PIL's Lanczos filter (what
  mediapy.resize_video does for uint8 videos) BEFORE the / 255 * 2 - 1 scaling: the scaled frames sit on the
  uint8 grid and equal a direct PIL resize."""
  import pickle
  from PIL import Image
  rng = np.random.default_rng(0)
  video = rng.integers(0, 256, (3, 40, 56, 3), dtype=np.uint8)
  pts = rng.uniform(0, 1, (5, 3, 2))
  occ = rng.random((5, 3)) < 0.3
  occ[:, 0] = False
  path = tmp_path / 'davis.pkl'
  with open(path, 'wb') as f:
    pickle.dump({'clip': dict(video=video, points=pts, occluded=occ)}, f)
  (name, ex), = list(tapvid.davis_examples(str(path), 'first', (32, 48)))
  frames = ex['video'][0]
  assert frames.shape == (3, 32, 48, 3) and frames.dtype == np.float32
  u8 = np.round((frames + 1.0) * 127.5)
  np.testing.assert_allclose((frames + 1.0) * 127.5, u8, atol=1e-3)          # on the uint8 grid
  ref = np.stack([np.asarray(Image.fromarray(f).resize((48, 32), resample=Image.Resampling.LANCZOS)) for f in video])
  np.testing.assert_array_equal(u8.astype(np.uint8), ref)
  # same size: untouched
  np.testing.assert_array_equal(tapvid.resize_video(video, (40, 56)), video)


def test_tracked_dataset_ground_truth_and_davis_layout(tmp_path):
  """synthetic.make_tracked_dataset (the offline AJ proxy's data): every track is a point of
  the moving texture -- the pixel under a visible track never changes -- occluded exactly while outside the
  frame, offsets in multiples of the backbone stride; written / read back in the TAP-Vid-DAVIS layout and
  scored by tapvid.evaluate with a tracker that returns the ground truth (AJ = 1)."""
  from tapnet_amd import synthetic
  T, S, N = 9, 64, 12
  data = synthetic.make_tracked_dataset(3, 2, T, S, S, N)
  assert sorted(data) == ['texture_00', 'texture_01']
  for d in data.values():
    assert d['video'].shape == (T, S, S, 3) and d['video'].dtype == np.uint8
    assert d['points'].shape == (N, T, 2) and d['occluded'].shape == (N, T)
    px = d['points'] * S
    inside = (px >= 0).all(-1) & (px < S).all(-1)
    np.testing.assert_array_equal(d['occluded'], ~inside)
    step = np.diff(px, axis=1)
    np.testing.assert_allclose(step, np.round(step / 8) * 8, atol=1e-9)     # multiples of the stride
    for n in range(N):
      vis = np.nonzero(~d['occluded'][n])[0]
      vals = np.stack([d['video'][t, int(px[n, t, 1]), int(px[n, t, 0])] for t in vis])
      assert (vals == vals[0]).all()
  path = str(tmp_path / 'tex.pkl')
  tapvid.write_davis_pickle(path, data)

  class Perfect:
    def __init__(self): self.ex = None
    def get_feature_grids(self, video): return None
    def __call__(self, video, is_training, qp, feature_grids=None):
      ex = self.ex
      idx = [int(np.argmin(np.abs(ex['query_points'][0] - q).sum(-1))) for q in qp[0]]
      occl = np.where(ex['occluded'][0][idx], 20.0, -20.0)
      out = dict(tracks=ex['target_points'][:, idx], occlusion=occl[None], expected_dist=np.full_like(occl, -20.0)[None])
      out.update({'unrefined_' + k: [v] for k, v in list(out.items())})
      return out

  for mode in ('strided', 'first'):
    exs = list(tapvid.davis_examples(path, mode, (S, S)))
    assert len(exs) == 2
    for it in (None, 0):
      m = Perfect()
      tot = []
      for name, ex in exs:
        m.ex = ex
        tot.append(tapvid.evaluate(m, [(name, ex)], query_mode=mode, iteration=it)['average_jaccard'])
      assert abs(np.mean(tot) - 1.0) < 1e-9
