"""CPU tests of the track-resident fused mixer kernel (tapnet_amd/csrc/mixer_fused.hpp) on the host
emulator (lane-accurate MFMA fragment layouts, shuffles, barriers): the packed weight streams, the
register-resident residual in MFMA accumulator layout, the DPP time shifts across token tiles, the
swizzled LDS activation images and ragged clip lengths -- against the numpy oracle (f32 build) and
against the separate-launch path on the same bf16-rounded operands (bf16 build)."""
import ctypes

import numpy as np
import pytest

from oracle import tapir_oracle as O
from tapnet_amd import _ffi, synthetic
from tests.emu_engine import EmuEngine


def _mixer(e, x, mode):
  assert e.lib.tapir_debug_set_mixer_mode(e.ctx, mode) == 0
  return e.pips_mixer(x)


@pytest.mark.parametrize('pyr,T,N', [(1, 19, 2), (0, 16, 1), (1, 33, 1), (0, 48, 1), (1, 5, 3)])
def test_fused_mixer_f32_vs_oracle(pyr, T, N):
  """f32 build (exact-f32 MFMA): 1, 2 and 3 token tiles, ragged T (masking of the padded
  tokens in both temporal convolutions), both input widths (486 -> 512, 535 -> 576)."""
  w = synthetic.make_weights(6 + T, pyr, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, pyramid_level=pyr, num_mixer_blocks=2, initial_resolution=(64, 64))
  rng = np.random.default_rng(T)
  x = rng.standard_normal((N, T, 388 + 49 * (2 + pyr))).astype(np.float32)
  out = _mixer(e, x, 2)
  ref, _ = O.pips_mlp_mixer(w, x, num_blocks=2)
  np.testing.assert_allclose(out, ref, atol=2e-4)
  sep = _mixer(e, x, 1)
  np.testing.assert_allclose(out, sep, atol=2e-4)
  e.close()


@pytest.mark.parametrize('T', [20, 48])
def test_fused_mixer_bf16_matches_separate_launches(T):
  """bf16 build: same roundings as the separate-launch path (LN output, GELU output and mixer input
  rounded to bf16, f32 residual / accumulation), so the two agree to accumulation-order noise;
  T = 48 is the benchmark's three-token-tile instantiation (unmasked), T = 20 a ragged two-tile one."""
  w = synthetic.make_weights(9, 1, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_mixer_blocks=2, initial_resolution=(64, 64), dtype=_ffi.TAPIR_BF16)
  rng = np.random.default_rng(T)
  x = rng.standard_normal((2, T, 535)).astype(np.float32)
  fused = _mixer(e, x, 2)
  sep = _mixer(e, x, 1)
  ref, _ = O.pips_mlp_mixer(w, x, num_blocks=2)
  assert np.abs(fused - sep).max() < 2e-2, np.abs(fused - sep).max()
  assert np.abs(fused - ref).max() < 0.15     # bf16 operand rounding vs the f32 oracle
  assert np.median(np.abs(fused - sep)) < 2e-3
  e.close()


def test_fused_mixer_rejects_unsupported_shapes():
  w = synthetic.make_weights(3, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_mixer_blocks=1, initial_resolution=(64, 64))
  assert e.lib.tapir_debug_set_mixer_mode(e.ctx, 2) == 0
  x = np.zeros((1, 49, 535), np.float32)    # the fused kernel covers up to 48 frames (3 token tiles)
  out = np.zeros((1, 49, 388), np.float32)
  p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
  rc = e.lib.tapir_pips_mixer(e.ctx, p(x), 1, 49, p(out), None, None, None, None, None)
  assert rc == _ffi.TAPIR_ERR_UNSUPPORTED
  e.close()


@pytest.mark.parametrize('T,N', [(48, 3), (40, 2), (20, 5), (70, 1), (96, 2)])
def test_wide_fused_mixer_bf16(T, N):
  """Wide form (mixer_fused_wide.hpp, mode 3): two tracks of up to 48 frames per workgroup (an odd
  track count leaves the last workgroup half empty; T = 40 / 20 are ragged) or one track of 49..96
  frames (5 and 6 token tiles); its own weight-stream packing (chunks of 256 hidden units, output
  Linear in two passes).  Same roundings as the separate-launch path."""
  w = synthetic.make_weights(9, 1, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_mixer_blocks=2, initial_resolution=(64, 64), dtype=_ffi.TAPIR_BF16)
  rng = np.random.default_rng(T + N)
  x = rng.standard_normal((N, T, 535)).astype(np.float32)
  wide = _mixer(e, x, 3)
  sep = _mixer(e, x, 1)
  assert np.isfinite(wide).all()
  assert np.abs(wide - sep).max() < 2e-2, np.abs(wide - sep).max()
  assert np.median(np.abs(wide - sep)) < 2e-3
  if T <= 48:
    narrow = _mixer(e, x, 2)
    assert np.abs(wide - narrow).max() < 2e-2
  e.close()


def test_wide_fused_mixer_needs_bf16():
  w = synthetic.make_weights(3, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_mixer_blocks=1, initial_resolution=(64, 64))   # f32 build
  assert e.lib.tapir_debug_set_mixer_mode(e.ctx, 3) == 0
  x = np.zeros((1, 48, 535), np.float32)
  out = np.zeros((1, 48, 388), np.float32)
  p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
  assert e.lib.tapir_pips_mixer(e.ctx, p(x), 1, 48, p(out), None, None, None, None, None) == _ffi.TAPIR_ERR_UNSUPPORTED
  e.close()


@pytest.mark.parametrize('mode,dtype,T,Q', [(2, _ffi.TAPIR_F32, 20, 3), (2, _ffi.TAPIR_BF16, 33, 2), (3, _ffi.TAPIR_BF16, 40, 3),
                                             (1, _ffi.TAPIR_F32, 9, 5), (1, _ffi.TAPIR_BF16, 11, 40)])
def test_fused_state_update_is_bit_identical(mode, dtype, T, Q):
  """refine_pips's state update (tapir_model.py:613-623, 1026-1039: pos / occ / expd / feats, the per-iteration
  output slices, the reset after a level) applied by the output stage of the track-resident mixer kernels
  (fused_emit) -- and, for few rows on the separate-launch path (mode 1), by the epilogue of the output Linear (gemm.hpp
  EPI_BIAS_UPDATE) -- against the separate update_kernel on the mixer's [R,388] output: the same operations in the same
  order, so every output of estimate_trajectories -- two refinement levels, first / later iterations -- is
  bit-identical; and (f32) equal to the oracle."""
  w = synthetic.make_weights(4, 1, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_pips_iter=2, num_mixer_blocks=2, initial_resolution=(64, 64), dtype=dtype)
  rng = np.random.default_rng(T)
  S = 64
  lows = [O.l2_normalize(rng.standard_normal((1, T, 8, 8, 256)).astype(np.float32)) for _ in range(2)]
  his = [O.l2_normalize(rng.standard_normal((1, T, 16, 16, 128)).astype(np.float32)) for _ in range(2)]
  lows, his, res = [lows[0], lows[0], lows[1]], [his[0], his[0], his[1]], [(S, S)] * 3
  qp = synthetic.make_queries(5, Q, T, S, S)
  ql, qh = O.get_query_features(lows, his, res, qp, (1, T, S, S, 3))
  assert e.lib.tapir_debug_set_mixer_mode(e.ctx, mode) == 0
  outs = {}
  for upd in (0, 1):
    assert e.lib.tapir_debug_set_update_mode(e.ctx, upd) == 0
    outs[upd] = e.estimate_trajectories((S, S), lows, his, res, ql, qh, qp)
  for k in ('tracks', 'occlusion', 'expected_dist'):
    np.testing.assert_array_equal(outs[0][k], outs[1][k])
  if dtype == _ffi.TAPIR_F32:
    ref = O.estimate_trajectories(w, (S, S), lows, his, res, ql, qh, qp, num_pips_iter=2, pyramid_level=1,
                                  initial_resolution=(S, S), num_blocks=2)
    for i in range(5):
      np.testing.assert_allclose(outs[1]['occlusion'][i], ref['occlusion'][i], atol=2e-4)
  e.close()


@pytest.mark.parametrize('dtype,T,Q,pyr', [(_ffi.TAPIR_F32, 20, 3, 1), (_ffi.TAPIR_BF16, 33, 2, 1), (_ffi.TAPIR_BF16, 48, 2, 0)])
def test_patch_rows_built_in_the_mixer_prologue_are_bit_identical(dtype, T, Q, pyr):
  """refine_pips's front half (tapir_model.py:496-594: header, features, 7x7 patch correlations per pyramid level)
  built by the track-resident mixer kernel in its prologue, straight into the swizzled LDS input image
  (mixer_fused.hpp fuse_patch, pips.hpp patch_row), against the separate patch_corr_kernel launch through HBM: the
  same device function, the same rounding to the operand type -- every output of estimate_trajectories over two
  refinement levels (first / later iterations: query features, then refined features) is bit-identical; ragged
  clips (20, 33 frames: zero rows past the clip end) and both pyramid depths."""
  w = synthetic.make_weights(6, pyr, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, pyramid_level=pyr, num_pips_iter=2, num_mixer_blocks=2, initial_resolution=(64, 64), dtype=dtype)
  rng = np.random.default_rng(T + pyr)
  S = 64
  lows = [O.l2_normalize(rng.standard_normal((1, T, 8, 8, 256)).astype(np.float32)) for _ in range(2)]
  his = [O.l2_normalize(rng.standard_normal((1, T, 16, 16, 128)).astype(np.float32)) for _ in range(2)]
  lows, his, res = [lows[0], lows[0], lows[1]], [his[0], his[0], his[1]], [(S, S)] * 3
  qp = synthetic.make_queries(7, Q, T, S, S)
  qp[0, 0, 1:] = [1.0, S - 0.5]        # a patch that leaves the grid
  ql, qh = O.get_query_features(lows, his, res, qp, (1, T, S, S, 3))
  assert e.lib.tapir_debug_set_mixer_mode(e.ctx, 2) == 0
  outs = {}
  for mode in (0, 1):
    assert e.lib.tapir_debug_set_patch_mode(e.ctx, mode) == 0
    outs[mode] = e.estimate_trajectories((S, S), lows, his, res, ql, qh, qp)
  for k in ('tracks', 'occlusion', 'expected_dist'):
    np.testing.assert_array_equal(outs[0][k], outs[1][k])
  assert np.isfinite(outs[1]['tracks']).all()
  e.close()


