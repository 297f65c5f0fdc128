"""`-m gpu`: parity AT THE BENCHMARKED CONFIGURATION (BASELINE.json configs[1]: 256x256x48 clip,
256 queries) -- the production kernel paths (persistent multi-tile GEMM walk, interior epilogues,
residual-as-accumulator across tiles, the fused mixer) against the numpy oracle.

The oracle is per-query independent (tapnet/tapvid/README.md:32-38; chunk loop
tapir_model.py:952), so comparing a SUBSET of the queries of the full-size call with the oracle
run on that subset alone is exact, and costs ~10 s of numpy per case instead of minutes.

Tolerances:
  * f32 build: 1e-3 abs (north_star) on tracks / occlusion / expected_dist, final and per
    iteration, gated on a clear argmax of the cost-volume heat map (soft-argmax is discontinuous
    at ties, model_utils.py:232).
  * bf16 build END TO END (bf16 backbone + bf16 hot path) vs the f32 oracle: the drift
    distribution is recorded (gpurun_out/accuracy_bf16.json -> profiles/) and gated; see
    test_bf16_end_to_end_vs_oracle.
"""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import tapir_oracle as O  # noqa: E402
from tapnet_amd import synthetic  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = {
    'tapir': dict(pyramid_level=0, extra_convs=False, softmax_temperature=20.0),
    'bootstapir': dict(pyramid_level=1, extra_convs=True, softmax_temperature=10.0),
}
T, S = 48, 256


def _clip():
  return synthetic.make_video(7, T, S, S)


def _np_grids(fg):
  return ([x.cpu().numpy() for x in fg.lowres], [x.cpu().numpy() for x in fg.hires],
          [tuple(r) for r in fg.resolutions])


def _oracle_subset(w, kw, video_shape, grids, qp, idx, rnd=None):
  lows, his, res = grids
  sub = qp[:, idx]
  ref = O.tapir_from_grids(w, video_shape, lows, his, res, sub, pyramid_level=kw['pyramid_level'],
                           softmax_temperature=kw['softmax_temperature'], rnd=rnd)
  ql, _ = O.get_query_features(lows, his, res, sub, video_shape)
  _, _, _, st = O.tracks_from_cost_volume(w, ql[0], lows[0], sub,
                                          softmax_temperature=kw['softmax_temperature'],
                                          return_stages=True)
  clear = (st['top2_rel_gap'] > 1e-4).all(axis=-1)   # [1, n]: every frame has a clear argmax
  return ref, clear


def _check_against(out, ref, idx, clear, atol):
  # queries set aside by the margin mask (a frame with a top-2 gap below 1e-4): observed 0 .. 2 of 16 on these
  # seeds (printed); the ungated checks at this shape are
  # tests/test_reference_headline_pin.py and tests/test_jax_reference_pin.py
  print(f'margin mask: {int(clear.sum())} of {clear.size} queries compared')
  assert clear.mean() >= 0.85, clear.mean()
  for k in ('tracks', 'occlusion', 'expected_dist'):
    np.testing.assert_allclose(out[k][:, idx][clear], ref[k][clear], atol=atol, err_msg=k)
  n_it = len(ref['unrefined_tracks'])
  assert len(out['unrefined_tracks']) == n_it
  for i in range(n_it):
    for k in ('unrefined_tracks', 'unrefined_occlusion', 'unrefined_expected_dist'):
      np.testing.assert_allclose(out[k][i][:, idx][clear], ref[k][i][clear], atol=atol,
                                 err_msg=f'{k}[{i}]')


@pytest.mark.parametrize('name,Q', [('tapir', 256), ('bootstapir', 256), ('tapir', 512)])
def test_f32_full_size_vs_oracle(name, Q):
  """f32 TAPIR.__call__ at 256x256x48 with Q = 256 (R = 12288 token rows: the config-2 tiles)
  and Q = 512 (R = 24576: the large-row tiles), TAPIR and BootsTAPIR kwargs, against
  O.tapir_from_grids on a 16-query subset: final outputs and every unrefined iteration."""
  from tapnet_amd import tapir_model
  kw = KW[name]
  w = synthetic.make_weights(3 if name == 'tapir' else 5, kw['pyramid_level'], kw['extra_convs'])
  video = _clip()
  qp = synthetic.make_queries(8 + Q, Q, T, S, S)
  m = tapir_model.TAPIR(**kw, weights=w, device='cuda:0')
  fg = m.get_feature_grids(torch.as_tensor(video).cuda())
  out = m(video, False, qp, feature_grids=fg)
  idx = np.random.default_rng(Q).choice(Q, 16, replace=False)
  ref, clear = _oracle_subset(w, kw, video.shape, _np_grids(fg), qp, idx)
  _check_against(out, ref, idx, clear, 1e-3)


def _stats(d):
  d = np.asarray(d, np.float64).ravel()
  return dict(median=float(np.median(d)), p90=float(np.percentile(d, 90)),
              p99=float(np.percentile(d, 99)), max=float(d.max()))


def _drift(out, ref, idx, clear):
  """drift statistics of a bf16 run against the f32 oracle on the query subset idx"""
  d0 = np.linalg.norm(out['unrefined_tracks'][0][:, idx] - ref['unrefined_tracks'][0], axis=-1)
  flip = d0 > 4.0                                   # another cell of the 8-px grid won the argmax
  keep = clear[..., None] & ~flip
  d = np.linalg.norm(out['tracks'][:, idx] - ref['tracks'], axis=-1)
  do = np.abs(out['occlusion'][:, idx] - ref['occlusion'])
  de = np.abs(out['expected_dist'][:, idx] - ref['expected_dist'])
  return dict(tracks_px=_stats(d[keep]), occlusion_logit=_stats(do[keep]),
              expected_dist_logit=_stats(de[keep]), argmax_flip_rate=float(flip[clear].mean()),
              tracks_px_incl_flips=_stats(d[clear]), points=int(keep.sum()))


def test_bf16_end_to_end_vs_oracle():
  """The configuration bench.py times -- bf16 backbone (bf16 activations through the ResNet, bf16
  MIOpen convolutions, bf16 glue kernels) + bf16 hot path -- against the f32 oracle fed with the
  f32 build's feature grids (themselves pinned to the reference by test_backbone_golden_gpu).

  Three measurements, recorded in gpurun_out/accuracy_bf16.json (copied to profiles/):
    hot_path : bf16 hot path on the F32 grids vs the oracle   (the HIP kernels' bf16 arithmetic)
    backbone : cosine of every pixel's bf16-backbone feature vector to the f32 one
    end_to_end: bf16 backbone + bf16 hot path vs the oracle   (what the bench runs)
  Drift = tracks (px) and occlusion / expected_dist logits outside argmax flips of the cost-volume
  initialisation, plus the flip rate (a different heat-map cell wins: an 8-px jump of
  unrefined_tracks[0]).

  What bounds it: with these random-init weights the refinement is not contractive -- the f32 HIP
  build differs from the f32 oracle by ~1e-4 px for ~1e-6 relative re-association noise (a x100
  amplification), so 2^-9 operand rounding gives tenths of a pixel; the SURVEY.md 7 estimate
  (median 8e-3 px) does not hold for them.  The hot-path gates are 2x the values measured on
  MI355X (profiles/r02_accuracy_bf16.json), the end-to-end ones 1.4-1.6x: regression gates, not an
  accuracy claim -- the stage-level claim is tests/test_gpu_bf16_stages.py (every bf16 kernel against the
  oracle with the bf16 build's roundings); AJ on TAP-Vid needs a trained checkpoint
  (tests/test_gpu_aj_proxy.py is the offline stand-in)."""
  from tapnet_amd import tapir_model
  kw = KW['tapir']
  w = synthetic.make_weights(3, kw['pyramid_level'], kw['extra_convs'])
  video = _clip()
  Q = 256
  qp = synthetic.make_queries(8 + Q, Q, T, S, S)
  m32 = tapir_model.TAPIR(**kw, weights=w, device='cuda:0')
  fg32 = m32.get_feature_grids(torch.as_tensor(video).cuda())
  idx = np.random.default_rng(1).choice(Q, 32, replace=False)
  ref, clear = _oracle_subset(w, kw, video.shape, _np_grids(fg32), qp, idx)
  out32 = m32(video, False, qp, feature_grids=fg32)
  d32 = np.linalg.norm(out32['tracks'][:, idx] - ref['tracks'], axis=-1)[clear]
  del m32
  m16 = tapir_model.TAPIR(**kw, weights=w, device='cuda:0', dtype='bfloat16')
  out_hot = m16(video, False, qp, feature_grids=fg32)
  hot = _drift(out_hot, ref, idx, clear)
  # the same run against the oracle WITH the bf16 build's operand roundings (oracle.tapir_oracle.bf16_round):
  # what is left is accumulation order and rounding-boundary flips, amplified by the refinement
  ref16, _ = _oracle_subset(w, kw, video.shape, _np_grids(fg32), qp, idx, rnd=O.bf16_round)
  hot16 = _drift(out_hot, ref16, idx, clear)
  e2e = _drift(m16(video, False, qp), ref, idx, clear)     # bf16 backbone inside
  fg16 = m16.get_feature_grids(torch.as_tensor(video).cuda())
  cos = {}
  for name, a, b in (('lowres', fg16.lowres[0], fg32.lowres[0]), ('hires', fg16.hires[0], fg32.hires[0])):
    c = (a * b).sum(-1).flatten().float().cpu().numpy()
    cos[name] = dict(min=float(c.min()), p01=float(np.percentile(c, 1)), median=float(np.median(c)))
  rec = dict(config='TAPIR kwargs, 256x256x48, Q=256 (32-query subset vs the f32 numpy oracle)',
             f32_build_tracks_px=_stats(d32), hot_path=hot, hot_path_vs_rounding_oracle=hot16,
             backbone_cosine=cos, end_to_end=e2e)
  os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
  with open(os.path.join(ROOT, 'gpurun_out', 'accuracy_bf16.json'), 'w') as f:
    json.dump(rec, f, indent=1)
  print(json.dumps(rec))
  assert rec['f32_build_tracks_px']['max'] <= 1e-3
  # gates = 2x the values measured on MI355X (profiles/r02_accuracy_bf16.json: hot path median 0.0155 px,
  # p99 0.137 px, flips 0.46 %; end to end 0.256 / 1.58 px, flips 9 %)
  assert hot['tracks_px']['median'] <= 0.031 and hot['tracks_px']['p99'] <= 0.28, rec
  assert hot['occlusion_logit']['p99'] <= 0.17 and hot['expected_dist_logit']['p99'] <= 0.15, rec
  assert hot['argmax_flip_rate'] <= 0.01, rec
  # against the rounding oracle the bf16 hot path must sit well inside its distance to the f32 oracle
  # (measured: median 0.0030 px, p99 0.0099, max 0.0155, no flips -- 5x / 14x closer than to the f32 oracle)
  assert hot16['tracks_px']['median'] <= 0.006 and hot16['tracks_px']['p99'] <= 0.02, rec
  assert hot16['tracks_px']['max'] <= 0.04 and hot16['argmax_flip_rate'] <= 0.005, rec
  assert e2e['tracks_px']['median'] <= 0.36 and e2e['tracks_px']['p99'] <= 2.6, rec
  assert e2e['occlusion_logit']['p99'] <= 0.5 and e2e['expected_dist_logit']['p99'] <= 0.5, rec
  assert e2e['argmax_flip_rate'] <= 0.12, rec
  assert cos['lowres']['min'] > 0.997 and cos['hires']['min'] > 0.999, rec


@pytest.mark.parametrize('tag,extra', [('tapir', False), ('boots', True)])
def test_bf16_backbone_golden(tag, extra):
  """bf16 Backbone.features (what bench.py runs) vs the reference's feature grids
  (tests/golden/backbone.npz).  The grids are unit vectors per pixel (components ~ 1/16);
  tolerance: every pixel's cosine to the reference > 0.998 and max abs component error 2.5e-2
  (measured on MI355X: min cosine 0.9986, max abs error 1.7e-2 with the extra convolutions; bf16 has
  8 mantissa bits and the activations pass 17-27 convolutions + norms in bf16)."""
  from tests.golden_util import GOLDEN_DIR
  from tapnet_amd import tapir_model
  g = np.load(os.path.join(GOLDEN_DIR, 'backbone.npz'))
  w = synthetic.make_weights(21, 1, extra)
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=extra, weights=w, device='cuda:0',
                        dtype='bfloat16')
  v = torch.as_tensor(g['video']).cuda()
  low, hi = m._backbone.features(v.reshape(-1, 64, 64, 3))
  for got, ref in ((low, g[f'{tag}_lowres'][0]), (hi, g[f'{tag}_hires'][0])):
    got = got.float().cpu().numpy()
    cos = (got * ref).sum(-1)
    err = np.abs(got - ref).max()
    print(tag, 'min cos', cos.min(), 'max abs err', err)
    assert cos.min() > 0.998, cos.min()
    assert err < 2.5e-2, err


def test_config5_shape_vs_oracle():
  """BASELINE configs[4] in shape (512x512 frames at 96 frames: two refinement levels = 8 iterations,
  64x64 / 128x128 grids at the 512 level, clips longer than the fused mixer covers -> large-row GEMM
  tiles + streamed token mixing, fused cost volume at level 0) with 512 queries, f32 build, against the
  oracle on an 8-query subset at 1e-3 -- the oracle comparison the round-1 config-5 run lacked."""
  from tapnet_amd import tapir_model
  kw = KW['bootstapir']
  w = synthetic.make_weights(5, kw['pyramid_level'], kw['extra_convs'])
  T5, S5, Q = 96, 512, 512
  video = synthetic.make_video(11, T5, S5, S5)
  qp = synthetic.make_queries(12, Q, T5, S5, S5)
  m = tapir_model.TAPIR(**kw, weights=w, device='cuda:0')
  fg = m.get_feature_grids(torch.as_tensor(video).cuda())
  assert [tuple(r) for r in fg.resolutions] == [(256, 256), (256, 256), (512, 512)]
  out = m(video, False, qp, feature_grids=fg)
  assert len(out['unrefined_tracks']) == 8
  idx = np.random.default_rng(5).choice(Q, 8, replace=False)
  ref, clear = _oracle_subset(w, kw, video.shape, _np_grids(fg), qp, idx)
  _check_against(out, ref, idx, clear, 1e-3)


def test_staged_grids_equal_the_cast_path():
  """bf16 build: inside TAPIR.__call__ the backbone's L2-normalise kernel writes the hot path's bf16 copies of the
  feature grids (row-major + the cost-volume kernel's tile order, tapir_l2_normalize_staged) and the hot path reads
  them (tapir_set_staged_grid) instead of casting the f32 grids itself.  Same values, same roundings: every output is
  bit-identical to the call that is handed the f32 grids -- eagerly (first calls) and from the replayed hipGraph,
  for TAPIR and BootsTAPIR kwargs (pooled pyramid level, ExtraConvs)."""
  from tapnet_amd import tapir_model
  for name in ('tapir', 'bootstapir'):
    kw = KW[name]
    w = synthetic.make_weights(5, kw['pyramid_level'], kw['extra_convs'])
    video = torch.as_tensor(_clip()).cuda()
    qp = torch.as_tensor(synthetic.make_queries(3, 64, T, S, S)).cuda()
    m = tapir_model.TAPIR(**kw, weights=w, device='cuda:0', dtype='bfloat16')
    fg = m.get_feature_grids(video)                       # f32 grids only: the hot path casts
    ref = m(video, False, qp, feature_grids=fg)
    for call in range(4):                                 # calls 0-1 eager, 2 captures, 3 replays the hipGraph
      out = m(video, False, qp)
      assert m._backbone.last_staged is not None, 'the backbone did not stage the operand copies'
      for k in ('tracks', 'occlusion', 'expected_dist'):
        assert torch.equal(out[k], ref[k]), (name, call, k)
      for a, b in zip(out['unrefined_tracks'], ref['unrefined_tracks']):
        assert torch.equal(a, b), (name, call)
    del m
