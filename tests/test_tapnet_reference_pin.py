"""the TAP-Net head pinned to the reference's own code.

tests/golden/tapnet_head.npz holds outputs of /root/reference's tapnet/models/tapnet_model.py
(TAPNet.__call__ :173-290 with feature_grid=, tracks_from_cost_volume :111-171) and model_utils.py executed
over numpy stand-ins for jax / haiku (oracle/hk_numpy_shim.py; generator oracle/make_tapnet_golden.py).
CPU: the numpy restatement (oracle.tapir_oracle.tapnet_tracks_from_cost_volume) and the emulated HIP kernel
(case c: num_heads = 2) against it; `-m gpu`: tapnet_amd.tapnet_model.TAPNet on the device.  The parameters are
stored in Haiku layout and go through the product's from_haiku_params."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_import
from oracle import tapir_oracle as O
from tapnet_amd import _ffi, tapnet_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'tapnet_head.npz'))


def _case(tag):
  params = {}
  for k in GOLD.files:
    if k.startswith(tag + '/params/'):
      mod, leaf = k[len(tag + '/params/'):].rsplit('/', 1)      # 'tap_net/~/cost_volume_regression_1', 'w'
      params.setdefault(mod, {})[leaf] = GOLD[k]
  assert all(m.startswith('tap_net/~/') for m in params)       # constructed in TAPNet.__init__
  w = tapnet_model.from_haiku_params(params)
  return (w, GOLD[tag + '/feature_grid'], GOLD[tag + '/query_points'],
          tuple(int(v) for v in GOLD[tag + '/video_shape']),
          {k: GOLD[f'{tag}/{k}'] for k in ('tracks', 'occlusion', 'query_feats')})


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_restatement_matches_the_reference_code(tag):
  w, grid, qp, shp, ref = _case(tag)
  assert w['tapnet_cost_volume_track_mods.hid1.weight'].shape[1] == int(GOLD[tag + '/num_heads'])
  ql, _ = O.get_query_features([grid], [grid[..., :128]], [shp[2:4]], qp, shp)
  np.testing.assert_allclose(ql[0], ref['query_feats'], atol=1e-6)            # (measured 2e-7)
  pts, occ, st = O.tapnet_tracks_from_cost_volume(w, ql[0], grid, qp, shp[2:4], return_stages=True)
  assert st['top2_rel_gap'].min() > 1e-3          # no near-tie in the fixtures: every point is comparable
  np.testing.assert_allclose(pts, ref['tracks'], atol=1e-4)                   # (measured 1.5e-5 px)
  np.testing.assert_allclose(occ, ref['occlusion'], atol=1e-6)                # (measured 6e-9)


@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_emulated_kernel_matches_the_reference_code(tag):
  from tests.emu_engine import EmuEngine
  w, grid, qp, shp, ref = _case(tag)
  e = EmuEngine(w, num_mixer_blocks=1, initial_resolution=shp[2:4], dtype=_ffi.TAPIR_F32)
  pts, occ = e.tapnet_tracks_from_cost_volume(ref['query_feats'], grid, qp)
  np.testing.assert_allclose(occ, ref['occlusion'], atol=1e-4)
  np.testing.assert_allclose(pts, ref['tracks'], atol=1e-3)
  e.close()


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['a', 'b', 'c'])
def test_gpu_head_matches_the_reference_code(tag):
  w, grid, qp, shp, ref = _case(tag)
  m = tapnet_model.TAPNet(num_heads=int(GOLD[tag + '/num_heads']), weights=w, device='cuda:0')
  out = m(shp, False, qp, query_chunk_size=4, get_query_feats=True, feature_grid=grid)
  np.testing.assert_allclose(out['query_feats'], ref['query_feats'], atol=2e-6)
  np.testing.assert_allclose(out['occlusion'], ref['occlusion'], atol=1e-4)
  np.testing.assert_allclose(out['tracks'], ref['tracks'], atol=1e-3)


@pytest.mark.skipif(not ref_import.reference_available(), reason='reference tree not present')
def test_committed_golden_regenerates_from_the_reference():
  """Re-runs the reference over the stand-ins in a subprocess (the stand-ins shadow `jax` in sys.modules)."""
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'make_tapnet_golden.py'), '--check'],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout + r.stderr
