"""Runs the UNMODIFIED HIP kernel sources on the CPU fiber emulator
(tests/hipemu) through the C ABI and checks them against the oracle and the
reference-generated golden fixtures.  This validates indexing / fragment
layouts / halo logic without a GPU; the `-m gpu` tests repeat the same checks
on the real gfx950 library."""
import ctypes

import numpy as np
import pytest

from oracle import tapir_oracle as O
from tapnet_amd import _ffi, synthetic
from tests.emu_engine import EmuEngine
from tests.golden_util import CASES, load_case


def bf16_round(x):
  u = np.ascontiguousarray(x, np.float32).view(np.uint32)
  u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
  return u.astype(np.uint32).view(np.float32)


@pytest.fixture(scope='module')
def small_engine():
  w = synthetic.make_weights(5, pyramid_level=1, extra_convs=False, num_mixer_blocks=2,
                             backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_mixer_blocks=2, initial_resolution=(64, 64))
  yield e, w
  e.close()


def test_build_cost_volume_f32(small_engine):
  e, _ = small_engine
  rng = np.random.default_rng(0)
  qf = rng.standard_normal((2, 37, 256)).astype(np.float32)
  grid = rng.standard_normal((2, 3, 8, 12, 256)).astype(np.float32)
  vol = e.build_cost_volume(qf, grid)
  ref = O.build_cost_volume(qf, grid).transpose(1, 2, 0, 3, 4)
  np.testing.assert_allclose(vol, ref, atol=2e-4)


def test_build_cost_volume_bf16():
  w = synthetic.make_weights(5, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_mixer_blocks=1, initial_resolution=(64, 64),
                dtype=_ffi.TAPIR_BF16)
  rng = np.random.default_rng(1)
  qf = rng.standard_normal((1, 130, 256)).astype(np.float32)
  grid = rng.standard_normal((1, 2, 8, 8, 256)).astype(np.float32)
  vol = e.build_cost_volume(qf, grid)
  ref = O.build_cost_volume(bf16_round(qf), bf16_round(grid)).transpose(1, 2, 0, 3, 4)
  np.testing.assert_allclose(vol, ref, atol=2e-3)
  e.close()


@pytest.mark.parametrize('name', ['tapir', 'bootstapir'])
def test_query_features_golden(name, small_engine):
  e, _ = small_engine
  cfg, g, _ = load_case(name)
  ql = e.get_query_features(g['lowres'][0], g['query_points'], g['video'].shape[2:4])
  qh = e.get_query_features(g['hires'][0], g['query_points'], g['video'].shape[2:4])
  np.testing.assert_allclose(ql, g['qlowres'][0], atol=2e-6)
  np.testing.assert_allclose(qh, g['qhires'][0], atol=2e-6)


@pytest.mark.parametrize('name', ['tapir', 'bootstapir'])
def test_cost_volume_stage_golden(name):
  cfg, g, w = load_case(name)
  e = EmuEngine(w, pyramid_level=cfg['pyramid_level'], softmax_temperature=cfg['softmax_temperature'],
                initial_resolution=(cfg['res'], cfg['res']))
  r = cfg['res'] / cfg['video']
  qp = g['query_points'] * np.array([1.0, r, r], np.float32)
  pts, occ, expd = e.tracks_from_cost_volume(g['qlowres'][0], g['lowres'][0], qp)
  _, _, _, st = O.tracks_from_cost_volume(w, g['qlowres'][0], g['lowres'][0], qp,
                                          (cfg['res'], cfg['res']), cfg['softmax_temperature'],
                                          return_stages=True)
  np.testing.assert_allclose(occ, g['cv_occ'], atol=1e-4)
  np.testing.assert_allclose(expd, g['cv_expd'], atol=1e-4)
  ok = st['top2_rel_gap'] > 1e-4
  np.testing.assert_allclose(pts[ok], g['cv_points'][ok], atol=1e-3)
  e.close()


def test_cost_volume_stage_odd_grid():
  """odd h/w exercises the XLA-SAME stride-2 padding (pad_lo = 1) and ragged tiles."""
  w = synthetic.make_weights(9, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, num_mixer_blocks=1, initial_resolution=(72, 40))
  rng = np.random.default_rng(3)
  grid = O.l2_normalize(rng.standard_normal((1, 4, 9, 5, 256)).astype(np.float32))
  qf = O.l2_normalize(rng.standard_normal((1, 7, 256)).astype(np.float32))
  qp = np.stack([rng.integers(0, 4, (1, 7)), rng.uniform(0, 72, (1, 7)),
                 rng.uniform(0, 40, (1, 7))], -1).astype(np.float32)
  pts, occ, expd = e.tracks_from_cost_volume(qf, grid, qp)
  rp, ro, re, st = O.tracks_from_cost_volume(w, qf, grid, qp, (72, 40), 20.0, return_stages=True)
  np.testing.assert_allclose(occ, ro, atol=1e-4)
  np.testing.assert_allclose(expd, re, atol=1e-4)
  ok = st['top2_rel_gap'] > 1e-4
  np.testing.assert_allclose(pts[ok], rp[ok], atol=1e-3)
  e.close()


@pytest.mark.parametrize('causal', [False, True])
def test_pips_mixer(causal):
  w = synthetic.make_weights(6, 1, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, num_mixer_blocks=2, use_causal_conv=causal, initial_resolution=(64, 64))
  rng = np.random.default_rng(2)
  x = rng.standard_normal((3, 19, 535)).astype(np.float32)   # T=19 -> time chunks with halos
  out = e.pips_mixer(x)
  ref, _ = O.pips_mlp_mixer(w, x, num_blocks=2, use_causal_conv=causal)
  np.testing.assert_allclose(out, ref, atol=2e-4)
  e.close()


def test_pips_mixer_causal_state():
  w = synthetic.make_weights(7, 1, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, num_mixer_blocks=2, use_causal_conv=True, initial_resolution=(64, 64))
  rng = np.random.default_rng(4)
  N = 3
  xs = rng.standard_normal((N, 3, 535)).astype(np.float32)
  c1 = np.zeros((2, N, 2, 512), np.float32); c2 = np.zeros((2, N, 2, 2048), np.float32)
  octx = {}
  for i in range(2):
    octx[f'block_{i}_causal_1'] = c1[i]; octx[f'block_{i}_causal_2'] = c2[i]
  for t in range(3):
    out, c1, c2 = e.pips_mixer(xs[:, t:t + 1], c1, c2, get_ctx=True)
    ref, octx = O.pips_mlp_mixer(w, xs[:, t:t + 1], 2, True, octx, True)
    np.testing.assert_allclose(out, ref, atol=2e-4)
    for i in range(2):
      np.testing.assert_allclose(c1[i], octx[f'block_{i}_causal_1'], atol=2e-4)
      np.testing.assert_allclose(c2[i], octx[f'block_{i}_causal_2'], atol=2e-4)
  e.close()


def test_refine_pips_vs_oracle(small_engine):
  """includes out-of-frame windows and the last_iter (per-frame query) path."""
  e, w = small_engine
  rng = np.random.default_rng(8)
  B, Q, T = 1, 5, 3
  hires = O.l2_normalize(rng.standard_normal((B, T, 16, 16, 128)).astype(np.float32))
  lowres = O.l2_normalize(rng.standard_normal((B, T, 8, 8, 256)).astype(np.float32))
  pyramid = [hires, lowres, O.avg_pool_2x2(lowres)]
  qh = rng.standard_normal((B, Q, 128)).astype(np.float32)
  ql = rng.standard_normal((B, Q, 256)).astype(np.float32)
  queries = [qh, ql, ql]
  pos = rng.uniform(-6, 70, (B, Q, T, 2)).astype(np.float32)
  occ = rng.standard_normal((B, Q, T)).astype(np.float32)
  expd = rng.standard_normal((B, Q, T)).astype(np.float32)
  for last in (None, rng.standard_normal((B, Q, T, 384)).astype(np.float32)):
    got = e.refine_pips(queries, pyramid, pos, occ, expd, last, (64, 64), (64, 64))
    ref = O.refine_pips(w, queries, pyramid, pos, occ, expd, (64, 64), last_iter=last,
                        resize_hw=(64, 64), num_blocks=2)
    for a, b in zip(got, ref[:4]):
      np.testing.assert_allclose(a, b, atol=3e-4)


def _run_estimate(e, cfg, g, **kw):
  return e.estimate_trajectories((cfg['video'], cfg['video']), g['lowres'], g['hires'],
                                 g['res_list'], g['qlowres'], g['qhires'], g['query_points'], **kw)


@pytest.mark.parametrize('name', ['tapir', 'bootstapir', 'multires'])
def test_estimate_trajectories_golden(name):
  """Whole hot path (R1) on the emulator vs the reference's outputs."""
  cfg, g, w = load_case(name)
  e = EmuEngine(w, pyramid_level=cfg['pyramid_level'], softmax_temperature=cfg['softmax_temperature'],
                initial_resolution=(cfg['res'], cfg['res']))
  out = _run_estimate(e, cfg, g)
  n_it = 4 * (len(g['res_list']) - 1)
  for i in range(n_it):
    np.testing.assert_allclose(out['tracks'][i], g[f'unrefined_tracks_{i}'], atol=1e-3)
    np.testing.assert_allclose(out['occlusion'][i], g[f'unrefined_occlusion_{i}'], atol=1e-3)
    np.testing.assert_allclose(out['expected_dist'][i], g[f'unrefined_expected_dist_{i}'], atol=1e-3)
  np.testing.assert_allclose(out['tracks'][4::4].mean(0), g['tracks'], atol=1e-3)
  np.testing.assert_allclose(out['occlusion'][4::4].mean(0), g['occlusion'], atol=1e-3)
  np.testing.assert_allclose(out['expected_dist'][4::4].mean(0), g['expected_dist'], atol=1e-3)
  e.close()


def test_causal_streaming_golden():
  """Online path: one frame per call with the causal state fed back (live_demo.py:62-77)."""
  cfg, g, w = load_case('causal')
  e = EmuEngine(w, pyramid_level=1, use_causal_conv=True, softmax_temperature=20.0,
                initial_resolution=(cfg['res'], cfg['res']))
  Q = cfg['Q']
  c1 = np.zeros((4, 12, Q, 2, 512), np.float32); c2 = np.zeros((4, 12, Q, 2, 2048), np.float32)
  tr, oc = [], []
  for t in range(cfg['T']):
    lo = [x[:, t:t + 1] for x in g['lowres']]; hi = [x[:, t:t + 1] for x in g['hires']]
    out = e.estimate_trajectories((cfg['video'], cfg['video']), lo, hi, g['res_list'],
                                  g['qlowres'], g['qhires'], None, ctx_in=(c1, c2), get_ctx=True)
    c1, c2 = out['ctx']
    tr.append(out['tracks'][-1]); oc.append(out['occlusion'][-1])
  np.testing.assert_allclose(np.concatenate(tr, 2), g['tracks'], atol=1e-3)
  np.testing.assert_allclose(np.concatenate(oc, 2), g['occlusion'], atol=1e-3)
  np.testing.assert_allclose(c1[-1, 0].reshape(1, Q, 2, 512), g['state_last_block_0_causal_1'], atol=1e-3)
  np.testing.assert_allclose(c2[-1, 11].reshape(1, Q, 2, 2048), g['state_last_block_11_causal_2'], atol=1e-3)
  e.close()


@pytest.mark.parametrize('tc', [0, 2, 3])
def test_mix_stream_kernel(tc):
  """The unrolled, persistent token-mixing kernel (one / several units per workgroup, odd and even
  walks over the two row buffers, partial last chunk) against the oracle's first half of a
  PIPsConvBlock."""
  import ctypes
  w = synthetic.make_weights(8, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, num_mixer_blocks=1, initial_resolution=(64, 64))
  rng = np.random.default_rng(tc)
  N, T = 3, 31
  x = rng.standard_normal((N, T, 512)).astype(np.float32)
  xo = np.zeros_like(x)
  xn = np.zeros((N * T, 512), np.float32)
  p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
  rc = e.lib.tapir_debug_mix(e.ctx, 0, p(x), p(xo), p(xn), N, T, tc, None)
  assert rc == 0
  # first half of PIPsConvBlock (tapir_model.py:111-121), from the oracle's own pieces
  pre = 'torch_pips_mixer.blocks.0.'
  y = O.layernorm(x, w[pre + 'layer_norm.weight'])
  y = O._depthwise_conv1d(y, w[pre + 'mlp1_up.weight'], w[pre + 'mlp1_up.bias'], 4, False)
  y = O.gelu_tanh(y)
  y = O._depthwise_conv1d(y, w[pre + 'mlp1_up_1.weight'], w[pre + 'mlp1_up_1.bias'], 1, False)
  ref_x = y[..., 0::4] + y[..., 1::4] + y[..., 2::4] + y[..., 3::4] + x
  ref_xn = O.layernorm(ref_x, w[pre + 'layer_norm_1.weight'])
  np.testing.assert_allclose(xo, ref_x, atol=2e-5)
  np.testing.assert_allclose(xn.reshape(N, T, 512), ref_xn, atol=2e-5)
  e.close()


def test_cost_volume_stage_bf16_mfma():
  """bf16 build: cost-volume heads with the occlusion convolution on the MFMAs
  (cv_heads_mfma_kernel) against the oracle on bf16-rounded operands; odd grid = ragged MFMA
  tiles and the pad_lo = 1 case of the stride-2 XLA SAME padding."""
  for (h, wd, seed) in ((8, 8, 11), (9, 5, 12)):
    w = synthetic.make_weights(9, 1, False, num_mixer_blocks=1, backbone=False)
    e = EmuEngine(w, num_mixer_blocks=1, initial_resolution=(8 * h, 8 * wd), dtype=_ffi.TAPIR_BF16)
    rng = np.random.default_rng(seed)
    grid = O.l2_normalize(rng.standard_normal((1, 4, h, wd, 256)).astype(np.float32))
    qf = O.l2_normalize(rng.standard_normal((1, 5, 256)).astype(np.float32))
    qp = np.stack([rng.integers(0, 4, (1, 5)), rng.uniform(0, 8 * h, (1, 5)),
                   rng.uniform(0, 8 * wd, (1, 5))], -1).astype(np.float32)
    pts, occ, expd = e.tracks_from_cost_volume(qf, grid, qp)
    rp, ro, re, st = O.tracks_from_cost_volume(w, bf16_round(qf), bf16_round(grid), qp,
                                               (8 * h, 8 * wd), 20.0, return_stages=True)
    np.testing.assert_allclose(occ, ro, atol=3e-2)     # hid1 / conv-3 weights rounded to bf16
    np.testing.assert_allclose(expd, re, atol=3e-2)
    ok = st['top2_rel_gap'] > 1e-3
    np.testing.assert_allclose(pts[ok], rp[ok], atol=2e-3)   # the soft-argmax path stays f32
    e.close()


@pytest.mark.parametrize('dtype', [_ffi.TAPIR_F32, _ffi.TAPIR_BF16])
def test_cost_volume_fused_matches_workspace_path(dtype):
  """The fused cost-volume kernel (costvol_fused.hpp: einsum on the matrix cores into LDS + heads, the
  two small convolutions as chained exact-f32 MFMA products) against the round-1 path (einsum GEMM into
  a workspace + heads kernel) on the full 32x32 grid: two clips, a query count that is not a multiple
  of the query tile, and against the oracle in the f32 build."""
  w = synthetic.make_weights(19, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, num_mixer_blocks=1, initial_resolution=(256, 256), dtype=dtype)
  rng = np.random.default_rng(5)
  B, Q, T = 2, 19, 2
  grid = O.l2_normalize(rng.standard_normal((B, T, 32, 32, 256)).astype(np.float32))
  qf = O.l2_normalize(rng.standard_normal((B, Q, 256)).astype(np.float32))
  qp = np.stack([rng.integers(0, T, (B, Q)), rng.uniform(0, 256, (B, Q)),
                 rng.uniform(0, 256, (B, Q))], -1).astype(np.float32)
  assert e.lib.tapir_debug_set_cv_mode(e.ctx, 0) == 0
  pts, occ, expd = e.tracks_from_cost_volume(qf, grid, qp)
  assert e.lib.tapir_debug_set_cv_mode(e.ctx, 1) == 0
  pts1, occ1, expd1 = e.tracks_from_cost_volume(qf, grid, qp)
  if dtype == _ffi.TAPIR_F32:
    rp, ro, re, st = O.tracks_from_cost_volume(w, qf, grid, qp, (256, 256), 20.0, return_stages=True)
    ok = st['top2_rel_gap'] > 1e-4
    np.testing.assert_allclose(occ, ro, atol=1e-4)
    np.testing.assert_allclose(expd, re, atol=1e-4)
    np.testing.assert_allclose(pts[ok], rp[ok], atol=1e-3)
    np.testing.assert_allclose(occ, occ1, atol=1e-4)
  else:
    # both paths round the operands of the einsum to bf16; the occlusion head sees hid1 rounded to
    # bf16 in both; the soft-argmax path is f32 in both
    np.testing.assert_allclose(occ, occ1, atol=2e-2)
    np.testing.assert_allclose(expd, expd1, atol=2e-2)
    d = np.linalg.norm(pts - pts1, axis=-1)
    assert np.median(d) < 1e-3 and np.mean(d < 0.05) > 0.97, (np.median(d), d.max())
  e.close()


@pytest.mark.parametrize('dtype,hw', [(_ffi.TAPIR_F32, (32, 32)), (_ffi.TAPIR_F32, (9, 5)), (_ffi.TAPIR_BF16, (16, 16))])
def test_tapnet_head_vs_oracle(dtype, hw):
  """TAPNet.tracks_from_cost_volume (tapnet_model.py:111-171) on the fused cost-volume
  kernel -- no ReLU after the stride-2 convolution, one occlusion logit, temperature 10 -- against the
  numpy restatement; a context that holds ONLY the TAP-Net head weights."""
  w = synthetic.make_tapnet_head_weights(3)
  h, wd = hw
  e = EmuEngine(w, num_mixer_blocks=1, initial_resolution=(8 * h, 8 * wd), dtype=dtype)
  rng = np.random.default_rng(11)
  B, Q, T = 1, 9, 3
  grid = O.l2_normalize(rng.standard_normal((B, T, h, wd, 256)).astype(np.float32))
  qf = O.l2_normalize(rng.standard_normal((B, Q, 256)).astype(np.float32))
  qp = np.stack([rng.integers(0, T, (B, Q)), rng.uniform(0, 8 * h, (B, Q)),
                 rng.uniform(0, 8 * wd, (B, Q))], -1).astype(np.float32)
  pts, occ = e.tapnet_tracks_from_cost_volume(qf, grid, qp)
  if dtype == _ffi.TAPIR_BF16:
    qf, grid = bf16_round(qf), bf16_round(grid)
  rp, ro, st = O.tapnet_tracks_from_cost_volume(w, qf, grid, qp, (8 * h, 8 * wd), return_stages=True)
  ok = st['top2_rel_gap'] > 1e-3
  np.testing.assert_allclose(occ, ro, atol=1e-4 if dtype == _ffi.TAPIR_F32 else 3e-2)
  np.testing.assert_allclose(pts[ok], rp[ok], atol=1e-3 if dtype == _ffi.TAPIR_F32 else 2e-3)
  # a TAP-Net-only context refuses the TAPIR entry points with an error code
  out = np.zeros((B, Q, T, 2), np.float32)
  rc = e.lib.tapir_tracks_from_cost_volume(e.ctx, qf.ctypes.data_as(ctypes.c_void_p), grid.ctypes.data_as(ctypes.c_void_p),
                                           None, B, Q, T, h, wd, out.ctypes.data_as(ctypes.c_void_p),
                                           out.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), None)
  assert rc == _ffi.TAPIR_ERR_WEIGHTS
  e.close()


@pytest.mark.parametrize('dtype', [_ffi.TAPIR_F32, _ffi.TAPIR_BF16])
def test_cost_volume_stage_when_frames_x_cells_is_not_a_multiple_of_four(dtype):
  """A 7 x 11 grid with 3 frames (231 values per query row): the fused kernel does not care, and the workspace path
  (cv_mode 1: einsum GEMM into a volume, then the heads) runs its GEMM over the next multiple of four with the grid's
  rows clamped and the volume's rows padded -- both against the oracle, and equal to each other."""
  w = synthetic.make_weights(3, 0, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, pyramid_level=0, num_mixer_blocks=1, initial_resolution=(56, 88), dtype=dtype)
  rng = np.random.default_rng(5)
  Q, T, h, wd = 6, 3, 7, 11
  grid = O.l2_normalize(rng.standard_normal((1, T, h, wd, 256)).astype(np.float32))
  qp = synthetic.make_queries(6, Q, T, 56, 88)
  qf, _ = O.get_query_features([grid], [grid[..., :128]], [(56, 88)], qp, (1, T, 56, 88, 3))
  g_ref, q_ref = (bf16_round(grid), bf16_round(qf[0])) if dtype == _ffi.TAPIR_BF16 else (grid, qf[0])
  rp, ro, re, st = O.tracks_from_cost_volume(w, q_ref, g_ref, qp, (56, 88), 20.0, return_stages=True)
  ok = st['top2_rel_gap'] > 1e-3
  outs = []
  for mode in (0, 1):
    assert e.lib.tapir_debug_set_cv_mode(e.ctx, mode) == 0
    pts, occ, expd = e.tracks_from_cost_volume(qf[0], grid, qp)
    np.testing.assert_allclose(occ, ro, atol=1e-4 if dtype == _ffi.TAPIR_F32 else 3e-2)
    np.testing.assert_allclose(pts[ok], rp[ok], atol=1e-3 if dtype == _ffi.TAPIR_F32 else 5e-3)
    outs.append(pts)
  np.testing.assert_allclose(outs[0][ok], outs[1][ok], atol=1e-3 if dtype == _ffi.TAPIR_F32 else 5e-3)
  e.close()


@pytest.mark.parametrize('dtype,hw,Q', [(_ffi.TAPIR_F32, (40, 48), 5), (_ffi.TAPIR_F32, (64, 64), 3), (_ffi.TAPIR_BF16, (35, 64), 7),
                                        (_ffi.TAPIR_BF16, (64, 33), 2)])
def test_cost_volume_rows_of_up_to_64_cells(dtype, hw, Q):
  """Round 4: the row-streamed cost-volume kernel on grids wider than 32 cells (`initial_resolution` up to 512 x 512;
  before: pixel-tiled kernel up to 34 x 34 padded cells, workspace path up to 1600 cells, nothing beyond): padded rows
  of 66, 6 maps x 6 waves (4 x 4 in the f32 build), three or four 16-pixel tiles per row, two tiles per
  occlusion-convolution row, the soft arg max streamed twice from the in-place logits.  Ragged everything: 40 x 48 and
  35 x 64 cells (odd height: pad_lo = 1 of the stride-2 window), 33 cells per row (a third tile with one pixel), query
  counts that fill no tile.  f32 against the oracle; bf16 against the oracle on bf16-rounded operands."""
  h, wd = hw
  w = synthetic.make_weights(23, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, num_mixer_blocks=1, initial_resolution=(8 * h, 8 * wd), dtype=dtype)
  rng = np.random.default_rng(h + wd)
  T = 2 if h * wd < 3000 else 1
  grid = O.l2_normalize(rng.standard_normal((1, T, h, wd, 256)).astype(np.float32))
  qf = O.l2_normalize(rng.standard_normal((1, Q, 256)).astype(np.float32))
  qp = np.stack([rng.integers(0, T, (1, Q)), rng.uniform(0, 8 * h, (1, Q)), rng.uniform(0, 8 * wd, (1, Q))], -1).astype(np.float32)
  pts, occ, expd = e.tracks_from_cost_volume(qf, grid, qp)
  if dtype == _ffi.TAPIR_F32:
    rp, ro, re, st = O.tracks_from_cost_volume(w, qf, grid, qp, (8 * h, 8 * wd), 20.0, return_stages=True)
    ok = st['top2_rel_gap'] > 1e-4
    np.testing.assert_allclose(occ, ro, atol=1e-4)
    np.testing.assert_allclose(expd, re, atol=1e-4)
    assert ok.mean() > 0.7
    np.testing.assert_allclose(pts[ok], rp[ok], atol=1e-3)
  else:
    rp, ro, re, st = O.tracks_from_cost_volume(w, bf16_round(qf), bf16_round(grid), qp, (8 * h, 8 * wd), 20.0, return_stages=True)
    np.testing.assert_allclose(occ, ro, atol=3e-2)     # hid1 / conv-3 weights rounded to bf16
    np.testing.assert_allclose(expd, re, atol=3e-2)
    ok = st['top2_rel_gap'] > 1e-3
    np.testing.assert_allclose(pts[ok], rp[ok], atol=2e-3)   # the soft-arg-max path stays f32
  e.close()
