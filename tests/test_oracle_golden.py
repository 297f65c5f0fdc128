"""Pins oracle/tapir_oracle.py to the REFERENCE via the committed fixtures.

The fixtures were produced by oracle/make_golden.py from the reference's own
PyTorch TAPIR (tapnet/torch/tapir_model.py) -- see that script.  Tolerances are
fp32 re-association noise between torch and numpy (1e-4 abs on logits-scale
tensors); tracks are additionally checked at the north_star tolerance 1e-3 px.
"""
import numpy as np
import pytest

from oracle import tapir_oracle as O
from tests.golden_util import CASES, load_case, oracle_kwargs

NONCAUSAL = [n for n, c in CASES.items() if not c['causal']]


@pytest.mark.parametrize('name', list(CASES))
def test_query_features(name):
  cfg, g, _ = load_case(name)
  ql, qh = O.get_query_features(g['lowres'], g['hires'], g['res_list'],
                                g['query_points'], g['video'].shape)
  for a, b in zip(ql, g['qlowres']):
    np.testing.assert_allclose(a, b, atol=2e-6)
  for a, b in zip(qh, g['qhires']):
    np.testing.assert_allclose(a, b, atol=2e-6)


@pytest.mark.parametrize('name', NONCAUSAL)
def test_cost_volume_stage(name):
  cfg, g, w = load_case(name)
  r = cfg['res'] / cfg['video']
  qp = g['query_points'] * np.array([1.0, r, r], np.float32)
  pts, occ, expd, st = O.tracks_from_cost_volume(
      w, g['qlowres'][0], g['lowres'][0], qp, (cfg['res'], cfg['res']),
      cfg['softmax_temperature'], return_stages=True)
  np.testing.assert_allclose(st['cost_volume'], g['cost_volume'], atol=2e-6)
  np.testing.assert_allclose(occ, g['cv_occ'], atol=1e-4)
  np.testing.assert_allclose(expd, g['cv_expd'], atol=1e-4)
  # soft-argmax is discontinuous at argmax ties (model_utils.py:232): compare
  # tracks only where the oracle's top-2 margin is far above fp32 noise.
  ok = st['top2_rel_gap'] > 1e-4
  assert ok.mean() > 0.95
  np.testing.assert_allclose(pts[ok], g['cv_points'][ok], atol=1e-3)


@pytest.mark.parametrize('name', NONCAUSAL)
def test_first_refinement(name):
  cfg, g, w = load_case(name)
  hw = (cfg['res'], cfg['res'])
  queries = [g['qhires'][1], g['qlowres'][1]]
  pyramid = [g['hires'][1], g['lowres'][1]]
  for _ in range(cfg['pyramid_level']):
    queries.append(queries[-1])
    pyramid.append(O.avg_pool_2x2(pyramid[-1]))
  out = O.refine_pips(w, queries, pyramid, g['cv_points'], g['cv_occ'], g['cv_expd'],
                      hw, last_iter=None, resize_hw=g['res_list'][1])
  np.testing.assert_allclose(out[0], g['it1_points'], atol=1e-3)
  np.testing.assert_allclose(out[1], g['it1_occ'], atol=1e-3)
  np.testing.assert_allclose(out[2], g['it1_expd'], atol=1e-3)
  np.testing.assert_allclose(out[3], g['it1_feats'], atol=1e-3)


@pytest.mark.parametrize('name', NONCAUSAL)
def test_full_call(name):
  cfg, g, w = load_case(name)
  out = O.tapir_from_grids(w, g['video'].shape, g['lowres'], g['hires'], g['res_list'],
                           g['query_points'], **oracle_kwargs(cfg))
  np.testing.assert_allclose(out['tracks'], g['tracks'], atol=1e-3)
  np.testing.assert_allclose(out['occlusion'], g['occlusion'], atol=1e-3)
  np.testing.assert_allclose(out['expected_dist'], g['expected_dist'], atol=1e-3)
  n_it = len(out['unrefined_tracks'])
  assert n_it == 4 * (len(g['res_list']) - 1)
  for i in range(n_it):
    np.testing.assert_allclose(out['unrefined_tracks'][i], g[f'unrefined_tracks_{i}'], atol=1e-3)
    np.testing.assert_allclose(out['unrefined_occlusion'][i], g[f'unrefined_occlusion_{i}'], atol=1e-3)


def test_chunk_invariance():
  """Per-query outputs must not depend on the chunking (tapnet/tapvid/README.md:32-38)."""
  cfg, g, w = load_case('tapir')
  kw = oracle_kwargs(cfg)
  a = O.tapir_from_grids(w, g['video'].shape, g['lowres'], g['hires'], g['res_list'],
                         g['query_points'], **kw)
  b = O.tapir_from_grids(w, g['video'].shape, g['lowres'], g['hires'], g['res_list'],
                         g['query_points'], query_chunk_size=3, **kw)
  np.testing.assert_allclose(a['tracks'], b['tracks'], atol=1e-3)


def test_causal_streaming():
  """Online path (live_demo.py:62-77): one frame at a time with causal state."""
  cfg, g, w = load_case('causal')
  kw = oracle_kwargs(cfg)
  kw.pop('num_pips_iter')
  ql, qh = g['qlowres'], g['qhires']
  state = O.construct_initial_causal_state(cfg['Q'], len(g['res_list']) - 1)
  tr, oc, ex = [], [], []
  for t in range(cfg['T']):
    lo = [x[:, t:t + 1] for x in g['lowres']]
    hi = [x[:, t:t + 1] for x in g['hires']]
    traj = O.estimate_trajectories(w, (cfg['video'], cfg['video']), lo, hi, g['res_list'],
                                   ql, qh, None, causal_context=state,
                                   get_causal_context=True, **kw)
    state = traj['causal_context']
    tr.append(traj['tracks'][-1]); oc.append(traj['occlusion'][-1]); ex.append(traj['expected_dist'][-1])
  np.testing.assert_allclose(np.concatenate(tr, 2), g['tracks'], atol=1e-3)
  np.testing.assert_allclose(np.concatenate(oc, 2), g['occlusion'], atol=1e-3)
  np.testing.assert_allclose(np.concatenate(ex, 2), g['expected_dist'], atol=1e-3)
  np.testing.assert_allclose(state[-1]['block_0_causal_1'], g['state_last_block_0_causal_1'], atol=1e-3)
  np.testing.assert_allclose(state[-1]['block_11_causal_2'], g['state_last_block_11_causal_2'], atol=1e-3)
  # invariant: streaming == whole-clip causal run with zero left padding
  whole = O.estimate_trajectories(w, (cfg['video'], cfg['video']), g['lowres'], g['hires'],
                                  g['res_list'], ql, qh, None, **kw)
  np.testing.assert_allclose(whole['tracks'][-1], g['tracks'], atol=1e-3)


def test_causal_streaming_with_point_update():
  """Online path with a mid-stream point replacement (update_query_features + zero causal state
  for the replaced points, tapir_model.py:1172-1203) on a 128x128 frame (two refinement levels,
  8 iterations of state), vs the reference torch twin's outputs (fixture causal_update)."""
  cfg, g, w = load_case('causal_update')
  kw = oracle_kwargs(cfg)
  kw.pop('num_pips_iter')
  ql, qh = list(g['qlowres']), list(g['qhires'])
  nres = len(g['res_list']) - 1
  assert nres == 2
  state = O.construct_initial_causal_state(cfg['Q'], nres)
  assert len(state) == 8
  tr, oc, ex = [], [], []
  vshape1 = (1, 1) + g['video'].shape[2:]
  for t in range(cfg['T']):
    lo = [x[:, t:t + 1] for x in g['lowres']]
    hi = [x[:, t:t + 1] for x in g['hires']]
    if t == cfg['update_frame']:
      nl, nh = O.get_query_features(lo, hi, g['res_list'], g['update_query_points'], vshape1)
      ql, qh, state = O.update_query_features(ql, qh, nl, nh, g['update_idx'], state)
      src = g['level_src']
      for i in sorted(set(int(s) for s in src)):
        np.testing.assert_allclose(ql[i], g[f'updated_qlowres_{i}'], atol=2e-6)
        np.testing.assert_allclose(qh[i], g[f'updated_qhires_{i}'], atol=2e-6)
      np.testing.assert_allclose(state[1]['block_3_causal_1'], g['state_after_update_block_3_causal_1'],
                                 atol=1e-3)
      assert not state[1]['block_3_causal_1'][:, g['update_idx']].any()
    traj = O.estimate_trajectories(w, (cfg['video'], cfg['video']), lo, hi, g['res_list'],
                                   ql, qh, None, causal_context=state,
                                   get_causal_context=True, **kw)
    state = traj['causal_context']
    tr.append(traj['tracks'][-1]); oc.append(traj['occlusion'][-1]); ex.append(traj['expected_dist'][-1])
  np.testing.assert_allclose(np.concatenate(tr, 2), g['tracks'], atol=1e-3)
  np.testing.assert_allclose(np.concatenate(oc, 2), g['occlusion'], atol=1e-3)
  np.testing.assert_allclose(np.concatenate(ex, 2), g['expected_dist'], atol=1e-3)
  np.testing.assert_allclose(state[-1]['block_0_causal_1'], g['state_last_block_0_causal_1'], atol=1e-3)
  np.testing.assert_allclose(state[-1]['block_11_causal_2'], g['state_last_block_11_causal_2'], atol=1e-3)
