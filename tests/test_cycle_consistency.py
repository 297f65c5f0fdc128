"""SURVEY.md 8 f4, second half: TAP-Net's forward-backward cycle-consistency tracker
(tapnet/training/supervised_point_prediction.py:443-546) -- einsum -> softmax -> heatmaps_to_points forward, the grid
sampled at the tracked points, the same soft arg max backward against the query's frame, occluded when the backward
point misses the query by more than 48 px.  Pinned like the TAP-Net head: the reference's OWN LINES :444-531 (lifted
out of the trainer's method, which cannot be imported) and its model_utils.py / transforms.py executed over the numpy
stand-ins (oracle/make_cycle_golden.py -> tests/golden/cycle_consistency.npz); line :537's batch-axis slice cannot
execute and is restated as intended (see the script).

CPU: the numpy restatement (oracle.tapir_oracle.cycle_consistency_tracks) and the emulated HIP path
(tapir_cycle_consistency_tracks: two head-less launches of the row-streamed cost-volume kernel, f32 and bf16 builds)
against the fixture; `reference` marker: the fixture regenerates from the reference tree.  GPU: the product's
TAPNet.cycle_consistency_tracks."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_import, tapir_oracle as O
from tapnet_amd import _ffi, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, 'tests', 'golden', 'cycle_consistency.npz'))
TAGS = ('a', 'b')


def _case(tag):
  return (G[f'{tag}_query_feats'], G[f'{tag}_grid'], G[f'{tag}_query_points'], tuple(int(v) for v in G[f'{tag}_im_hw']))


def _check(tag, tracks, occ, inv, atol):
  np.testing.assert_allclose(tracks, G[f'{tag}_tracks'], atol=atol)
  np.testing.assert_allclose(inv, G[f'{tag}_inverse_tracks'], atol=atol)
  # the verdict is binary: compare where the backward distance is not within 2 * atol of the 48-px threshold
  clear = np.abs(np.sqrt(G[f'{tag}_dist']) - 48.0) > 2 * atol
  assert clear.mean() > 0.9 and np.array_equal(occ[clear], G[f'{tag}_occlusion'][clear])
  assert set(np.unique(occ)) <= {-10.0, 10.0}


@pytest.mark.parametrize('tag', TAGS)
def test_oracle_matches_the_reference_lines(tag):
  qf, grid, qp, hw = _case(tag)
  tracks, occ, st = O.cycle_consistency_tracks(qf, grid, qp, hw, 10.0, 48.0, return_stages=True)
  _check(tag, tracks, occ, st['inverse_tracks'], 1e-4)


@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('dtype', [_ffi.TAPIR_F32, _ffi.TAPIR_BF16])
def test_emulated_kernels_match_the_reference_lines(tag, dtype):
  from tests.emu_engine import EmuEngine
  qf, grid, qp, hw = _case(tag)
  w = synthetic.make_weights(2, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, num_mixer_blocks=1, initial_resolution=hw, dtype=dtype)
  tracks, occ, inv = e.cycle_consistency_tracks(qf, grid, qp, hw)
  if dtype == _ffi.TAPIR_F32:
    _check(tag, tracks, occ, inv, 1e-3)
  else:   # operands of both contractions rounded to bf16: against the restatement with the same roundings
    rt, ro, st = O.cycle_consistency_tracks(qf, grid, qp, hw, 10.0, 48.0, rnd=O.bf16_round, return_stages=True)
    d = np.linalg.norm(tracks - rt, axis=-1)
    assert np.median(d) < 1e-3 and np.mean(d < 0.05) > 0.95, (np.median(d), d.max())
    assert np.mean(occ == ro) > 0.9
  e.close()


@pytest.mark.reference
@pytest.mark.skipif(not ref_import.reference_available() or ref_import.reference_is_staged_copy(),
                    reason='reference tree not present')
def test_fixture_regenerates_from_the_reference_lines():
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'make_cycle_golden.py'), '--check'],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and 'max |diff| 0.000e+00' in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize('tag', TAGS)
@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_gpu_cycle_consistency_matches_the_reference_lines(tag, dtype):
  import torch
  from tapnet_amd import tapnet_model
  qf, grid, qp, hw = _case(tag)
  m = tapnet_model.TAPNet(device='cuda:0', dtype=dtype)
  B, T = grid.shape[:2]
  tracks, occ, inv = m.cycle_consistency_tracks(qf, grid, qp, (B, T, hw[0], hw[1], 3), return_inverse_tracks=True)
  if dtype == 'float32':
    _check(tag, tracks, occ, inv, 1e-3)
  else:
    rt, ro, _ = O.cycle_consistency_tracks(qf, grid, qp, hw, 10.0, 48.0, rnd=O.bf16_round, return_stages=True)
    d = np.linalg.norm(tracks - rt, axis=-1)
    assert np.median(d) < 1e-3 and np.mean(d < 0.05) > 0.95, (np.median(d), d.max())
    assert np.mean(occ == ro) > 0.9
