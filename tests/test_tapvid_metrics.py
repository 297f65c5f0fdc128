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
