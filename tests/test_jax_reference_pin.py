"""The reference's JAX / Haiku text executed over numpy stand-ins (oracle/hk_numpy_shim.py; generator
oracle/make_jax_golden.py -> tests/golden/jax_*.npz) against the oracle (CPU) and the engine (`-m gpu`).

What these fixtures add to the torch-twin fixtures of oracle/make_golden.py: NON-SQUARE clips
(tapnet/torch/utils.py:104 normalises both sampling axes by the height, the JAX text is per axis), the
antialiased down-resize of the multi-resolution path (tapir_model.py:670), the Haiku parameter tree
(tapnet_amd.weights.{torch_to_haiku_names, haiku_to_torch_names}: the generator checks the converted tree
against the tree the reference's own modules create, and ParameterizedTAPIR.apply looks every leaf up by name),
and the causal-state keys (tapir_model.py:1157-1170).  Weights are synthetic.make_weights(seed)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import backbone_torch, jax_resize, ref_import
from oracle import tapir_oracle as O
from tapnet_amd import synthetic, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIPS = ['tapir_nonsquare', 'bootstapir_multires', 'batch2_chunked']
ONLINE = ['causal_online', 'causal_update']


def _load(name):
  g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', f'jax_{name}.npz')))
  from oracle import make_jax_golden as gen       # the case table only; nothing of the reference is imported
  c = gen.CASES[name]
  kw = dict(c['kw'])
  assert kw['pyramid_level'] == int(g['pyramid_level']) and int(g['seed']) == c['seed']
  w = synthetic.make_weights(c['seed'], kw['pyramid_level'], kw['extra_convs'])
  g['case'] = c
  return g, kw, w


def _levels(g):
  n = sum(1 for k in g if k.startswith('lowres_') and k[7:].isdigit())
  return ([g[f'lowres_{i}'] for i in range(n)], [g[f'hires_{i}'] for i in range(n)],
          [tuple(int(v) for v in g[f'resolution_{i}']) for i in range(n)])


def _features(w, extra, frames):
  """Plain-PyTorch backbone restatement, frame by frame semantics (InstanceNorm: per frame)."""
  b, t, h, wd, _ = frames.shape
  lo, hi = backbone_torch.TorchBackbone(w, extra).features(torch.as_tensor(frames).reshape(-1, h, wd, 3))
  return lo.numpy().reshape(b, t, *lo.shape[1:]), hi.numpy().reshape(b, t, *hi.shape[1:])


@pytest.mark.parametrize('name', CLIPS)
def test_oracle_matches_the_jax_text(name):
  g, kw, w = _load(name)
  lows, his, res = _levels(g)
  video = g['video']
  assert res[-1] == video.shape[2:4] and res[0] == kw['initial_resolution']
  # backbone (plain-PyTorch restatement) at the clip's own resolution and, for the multi-resolution clip, on the
  # antialiased down-resize (oracle/jax_resize.py: the same restatement the stand-in resizes with)
  lo, hi = _features(w, kw['extra_convs'], video)
  np.testing.assert_allclose(lo, lows[-1], atol=2e-5)                      # (measured 9e-7)
  np.testing.assert_allclose(hi, his[-1], atol=2e-5)
  if res[0] != res[-1]:
    lo0, hi0 = _features(w, kw['extra_convs'], jax_resize.resize_bilinear(video, res[0]))
    np.testing.assert_allclose(lo0, lows[0], atol=2e-5)
    np.testing.assert_allclose(hi0, his[0], atol=2e-5)
  out = O.tapir_from_grids(w, video.shape, lows, his, res, g['query_points'], pyramid_level=kw['pyramid_level'],
                           softmax_temperature=20.0, initial_resolution=kw['initial_resolution'])
  for k in ('tracks', 'occlusion', 'expected_dist'):
    np.testing.assert_allclose(out[k], g[k], atol=1e-3 if k == 'tracks' else 1e-4)   # (measured 4e-5 px / 3e-6)
  assert len(out['unrefined_tracks']) == 4 * (len(res) - 1)
  for i, (t, o) in enumerate(zip(out['unrefined_tracks'], out['unrefined_occlusion'])):
    np.testing.assert_allclose(t, g['unrefined_tracks'][i], atol=1e-3)
    np.testing.assert_allclose(o, g['unrefined_occlusion'][i], atol=1e-4)


@pytest.mark.parametrize('name', ONLINE)
def test_oracle_online_matches_the_jax_text(name):
  """tapnet/live_demo.py:51-77 driven through the JAX text: features of frame 0, then one frame at a time;
  causal_update replaces two points mid-stream (update_query_features with the causal state, :1172-1203)."""
  g, kw, w = _load(name)
  c = g['case']
  video, qp = g['video'], g['query_points']
  res = O.generate_default_resolutions(video.shape[2:4], kw['initial_resolution'])
  res = [kw['initial_resolution']] + [tuple(r) for r in res]
  lo, hi = _features(w, kw['extra_convs'], video)
  f1 = video[:, :1].shape
  ql, qh = O.get_query_features([lo[:, :1]] * len(res), [hi[:, :1]] * len(res), res, qp, f1)
  state = O.construct_initial_causal_state(qp.shape[1], len(res) - 1)
  tr, oc, ex = [], [], []
  for t in range(video.shape[1]):
    lt, ht = [lo[:, t:t + 1]] * len(res), [hi[:, t:t + 1]] * len(res)
    if t == c.get('update_frame', -1):
      nl, nh = O.get_query_features(lt, ht, res, g['new_query_points'], f1)
      ql, qh, state = O.update_query_features(ql, qh, nl, nh, c['update_idx'], state)
    traj = O.estimate_trajectories(w, video.shape[2:4], lt, ht, res, ql, qh, None, causal_context=state,
                                   get_causal_context=True, pyramid_level=kw['pyramid_level'],
                                   softmax_temperature=20.0, initial_resolution=kw['initial_resolution'],
                                   use_causal_conv=True)
    state = traj['causal_context']
    tr.append(traj['tracks'][-1]); oc.append(traj['occlusion'][-1]); ex.append(traj['expected_dist'][-1])
  np.testing.assert_allclose(ql[0], g['query_lowres_0'], atol=2e-5)       # the query features after the updates
  np.testing.assert_allclose(np.concatenate(tr, 2), g['tracks'], atol=1e-3)
  np.testing.assert_allclose(np.concatenate(oc, 2), g['occlusion'], atol=1e-4)
  np.testing.assert_allclose(np.concatenate(ex, 2), g['expected_dist'], atol=1e-4)
  # the oracle's (torch-twin) state keys block_{i}_causal_{j} against the Haiku keys of the JAX text
  np.testing.assert_allclose(state[-1]['block_0_causal_1'], g['state_block_causal_1'], atol=1e-4)
  np.testing.assert_allclose(state[-1]['block_11_causal_2'], g['state_block_11_causal_2'], atol=1e-4)


@pytest.mark.parametrize('pyr,extra', [(0, False), (1, True)])
def test_haiku_tree_round_trip(pyr, extra):
  w = synthetic.make_weights(3, pyr, extra, num_mixer_blocks=3)
  hk = weights.torch_to_haiku_names(w)
  assert weights.is_haiku_params(hk)
  assert hk['tapir/~/resnet/~/block_group_2/~/block_0/~/shortcut_conv']['w'].shape == (1, 1, 128, 256)
  assert hk['tapir/~/pips_mlp_mixer/block_2/mlp1_up_1']['w'].shape == (3, 1, 2048)
  assert ('tapir/~/extra_convs/conv2_d_9' in hk) == extra
  back = weights.haiku_to_torch_names(hk)
  assert set(back) == set(w)
  for k in w:
    np.testing.assert_array_equal(back[k], w[k])


@pytest.mark.skipif(not ref_import.reference_available(), reason='reference tree not present')
def test_committed_goldens_regenerate_from_the_reference():
  """Re-runs the reference's JAX text over the stand-ins in a subprocess (they shadow `jax` in sys.modules);
  this is also where the Haiku tree of the converter is held to the tree the reference's modules create.
  Two cases by default (25 s); TAPNET_FULL_REGEN=1 re-runs all of them and the robotap driver (2.5 min)."""
  full = os.environ.get('TAPNET_FULL_REGEN') == '1'
  cases = [] if full else ['tapir_nonsquare', 'causal_online']
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'make_jax_golden.py'), '--check'] + cases,
                     capture_output=True, text=True, timeout=1800)
  assert r.returncode == 0, r.stdout + r.stderr
  assert r.stdout.count('leaves match') == (5 if full else 2)
  assert not full or '[robotap] max' in r.stdout


# ----------------------------------------------------------------------------------------------- engine
def _np(x):
  return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def _model(kw, w):
  from tapnet_amd import tapir_model
  # the reference's entry point: ParameterizedTAPIR(params, state, tapir_kwargs) on a HAIKU tree
  return tapir_model.ParameterizedTAPIR(weights.torch_to_haiku_names(w), None, tapir_kwargs=kw, device='cuda:0')


@pytest.mark.gpu
@pytest.mark.parametrize('name', CLIPS)
def test_gpu_matches_the_jax_text(name):
  g, kw, w = _load(name)
  m = _model(kw, w)
  lows, his, res = _levels(g)
  fg = m.get_feature_grids(g['video'], False)
  assert [tuple(r.shape[:2]) for r in fg.resolutions] == res
  for i in range(len(res)):
    np.testing.assert_allclose(_np(fg.lowres[i]), lows[i], atol=5e-4)
    np.testing.assert_allclose(_np(fg.hires[i]), his[i], atol=5e-4)
  out = m(g['video'], False, g['query_points'], query_chunk_size=int(g['query_chunk_size']))
  for k in ('tracks', 'occlusion', 'expected_dist'):
    np.testing.assert_allclose(_np(out[k]), g[k], atol=1e-3, err_msg=k)
  for i in range(len(out['unrefined_tracks'])):
    np.testing.assert_allclose(_np(out['unrefined_tracks'][i]), g['unrefined_tracks'][i], atol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ONLINE)
def test_gpu_online_matches_the_jax_text(name):
  g, kw, w = _load(name)
  c = g['case']
  m = _model(kw, w)
  video, qp = g['video'], g['query_points']
  qf = m.get_query_features(video[:, :1], False, qp)
  state = m.construct_initial_causal_state(qp.shape[1], len(qf.resolutions) - 1)
  assert sorted(state[0]) == sorted(str(k) for k in g['causal_state_keys'])      # Haiku keys for a Haiku tree
  tr, oc, ex = [], [], []
  for t in range(video.shape[1]):
    fg = m.get_feature_grids(video[:, t:t + 1], False)
    if t == c.get('update_frame', -1):
      new_qf = m.get_query_features(video[:, t:t + 1], False, g['new_query_points'], feature_grids=fg)
      qf, state = m.update_query_features(qf, new_qf, tuple(c['update_idx']), state)
    traj = m.estimate_trajectories(video.shape[2:4], False, fg, qf, None, query_chunk_size=c['chunk'],
                                   causal_context=state, get_causal_context=True)
    state = traj['causal_context']
    tr.append(_np(traj['tracks'][-1])); oc.append(_np(traj['occlusion'][-1]))
    ex.append(_np(traj['expected_dist'][-1]))
  np.testing.assert_allclose(np.concatenate(tr, 2), g['tracks'], atol=1e-3)
  np.testing.assert_allclose(np.concatenate(oc, 2), g['occlusion'], atol=1e-3)
  np.testing.assert_allclose(np.concatenate(ex, 2), g['expected_dist'], atol=1e-3)
  np.testing.assert_allclose(_np(state[-1]['tapir/~/pips_mlp_mixer/block_11_causal_2']),
                             g['state_block_11_causal_2'], atol=1e-3)


# ------------------------------------------------------------------------- robotap bulk tracking (8f row 3)
def _robotap():
  from oracle import make_jax_golden as gen
  return gen, dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'jax_robotap.npz')))


def test_bulk_sampling_and_checkpoint_file_match_the_reference_driver(tmp_path):
  """tests/golden/jax_robotap.npz = the reference's OWN track_many_points (tapir_clustering.py:1023-1179) run over
  the stand-ins on a Haiku checkpoint FILE: the sampled query points of tapnet_amd.bulk_tracking (draw order of
  np.random.seed(42)) and the .npy reader of tapnet_amd.weights on the same file."""
  from tapnet_amd import bulk_tracking as bt
  gen, g = _robotap()
  c = gen.ROBOTAP
  videos = gen.robotap_videos()
  s = bt.sample_query_points([v.shape for v in videos.values()], c['frame_stride'], c['points_per_frame'])
  np.testing.assert_array_equal(np.concatenate([np.full(len(yx), v) for v, _, yx in s]), g['query_points_0'])
  np.testing.assert_array_equal(np.concatenate([np.full(len(yx), i) for _, i, yx in s]), g['query_points_1'])
  np.testing.assert_allclose(np.concatenate([yx for _, _, yx in s]), g['query_points_2'], atol=1e-12)
  path = str(tmp_path / 'causal_tapir_checkpoint.npy')
  gen.robotap_checkpoint(path)
  w = weights.load_checkpoint(path)
  ref = synthetic.make_weights(c['seed'], 1, False)
  assert set(w) == set(ref)
  for k in ref:
    np.testing.assert_array_equal(w[k], ref[k])


@pytest.mark.gpu
def test_gpu_track_many_points_matches_the_reference_driver(tmp_path):
  from tapnet_amd import bulk_tracking as bt
  from tapnet_amd import tapir_model
  gen, g = _robotap()
  c = gen.ROBOTAP
  videos = gen.robotap_videos()
  path = str(tmp_path / 'causal_tapir_checkpoint.npy')
  gen.robotap_checkpoint(path)
  model = tapir_model.TAPIR(use_causal_conv=True, weights=weights.load_checkpoint(path), device='cuda:0')
  res = bt.track_many_points(videos, list(videos), model, frame_stride=c['frame_stride'],
                             points_per_frame=c['points_per_frame'], point_batch_size=c['point_batch_size'])
  np.testing.assert_allclose(res['query_features'].lowres[0], g['query_lowres_0'], atol=5e-4)
  for k in videos:
    assert res['separation_tracks'][k].shape == g[f'tracks_{k}'].shape
    np.testing.assert_allclose(res['separation_tracks'][k], g[f'tracks_{k}'], atol=2e-3)
    assert np.mean(res['separation_visibility'][k] == g[f'visibility_{k}']) == 1.0
  for i in range(3):
    np.testing.assert_allclose(res['query_points'][i], g[f'query_points_{i}'], atol=1e-12)


# ------------------------------------------------------------------------- the north_star shape
@pytest.mark.gpu
def test_gpu_north_star_shape_matches_the_jax_text():
  """BASELINE.json configs[1] -- 48 frames of 256x256, 256 queries, TAPIR kwargs, query chunks of 64 -- through the
  reference's JAX text over the stand-ins (tests/golden/jax_north_star.npz: outputs only; clip, queries and weights
  are seeds) against the f32 engine loaded from the Haiku tree: video -> tracks at 1e-3, every refinement iteration."""
  from oracle import make_jax_golden as gen
  g, kw, w = _load('north_star')
  video, qp = gen.case_inputs(g['case'])
  m = _model(kw, w)
  out = m(video, False, qp, query_chunk_size=int(g['query_chunk_size']))
  for k in ('tracks', 'occlusion', 'expected_dist'):
    d = np.abs(_np(out[k]) - g[k])
    print(k, 'max', d.max(), 'median', np.median(d))
    assert d.max() < 1e-3, (k, d.max())
  for i in range(len(out['unrefined_tracks'])):
    np.testing.assert_allclose(_np(out['unrefined_tracks'][i]), g['unrefined_tracks'][i], atol=1e-3)
    np.testing.assert_allclose(_np(out['unrefined_occlusion'][i]), g['unrefined_occlusion'][i], atol=1e-3)
