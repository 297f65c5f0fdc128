"""The few-frame form of the block convolutions (csrc/conv_small.hpp: a workgroup per (tile of whole rows, 64 output
channels), its four waves splitting the taps, partial tiles meeting in LDS) on the fiber emulator: against numpy on the same
bf16-rounded operands (resnet.py:241-256), against the many-frame kernel (csrc/conv_fused.hpp: same values up to the summation
order), its tile summaries against numpy, and the in-launch merge of the next norm's (a, b) pairs against the consuming
call's own merge launch.  Selected per clip by tapir_conv_set_small (tapnet_amd.backbone: clips of fewer than 4 frames -- the
online model, tapnet/live_demo.py:51-77)."""
import ctypes

import numpy as np
import pytest

from tapnet_amd import _ffi
from tests.emu_engine import emu_lib
from tests.test_conv_fused_emulated import _conv_ref, _ctx, _p, _r
from tests.test_gemm_tiles_emulated import from_bf16_bits, to_bf16_bits


def _norm_pairs(x, gamma, beta):
  mean = x.mean((1, 2), keepdims=True, dtype=np.float64)
  rstd = 1.0 / np.sqrt(x.astype(np.float64).var((1, 2), keepdims=True) + 1e-5)
  return (rstd * gamma).astype(np.float32), (beta - mean * rstd * gamma).astype(np.float32)


@pytest.mark.parametrize('cin,cout,ks,stride,H,W,shortcut', [
    (256, 256, 3, 1, 6, 32, True),      # the 32 x 32 maps' geometry: 2 rows = 64 pixels per tile, 4 channel groups
    (256, 256, 3, 1, 5, 16, False),     # 4 rows per tile, a ragged last tile
    (128, 128, 3, 1, 3, 64, True),      # one 64-pixel row per tile
    (64, 64, 3, 1, 2, 128, True),       # one 128-pixel row per tile: 8 fragments per wave
    (64, 64, 3, 1, 3, 40, False),       # 40-pixel rows: a masked fragment
    (64, 128, 3, 2, 6, 20, False),      # conv_0 of a stride-2 group
    (128, 256, 3, 2, 7, 12, False),
    (64, 128, 1, 2, 6, 20, False),      # proj_conv: two k-steps, waves 2 and 3 have none
    (256, 256, 1, 1, 4, 32, False),
])
def test_few_frame_form_against_numpy_and_the_many_frame_kernel(cin, cout, ks, stride, H, W, shortcut):
  lib = emu_lib()
  ctx = _ctx(lib)
  rng = np.random.default_rng(cin + 7 * H + W + ks)
  N = 2
  x = _r(rng.standard_normal((N, H, W, cin)) * 1.5 + 0.5)
  w = (rng.standard_normal((cout, cin, ks, ks)) / np.sqrt(ks * ks * cin)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, cin).astype(np.float32)
  beta = (rng.standard_normal(cin) * 0.3).astype(np.float32)
  Ho, Wo = -(-H // stride), -(-W // stride)
  sc = _r(rng.standard_normal((N, Ho, Wo, cout))) if shortcut else None
  xb = to_bf16_bits(x)
  part_in = np.zeros((N, 2, cin, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, cin, 2, None) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(np.ascontiguousarray(w)), cout, cin, ks, ctypes.byref(ws)) == 0

  def run(small):
    assert lib.tapir_conv_set_small(ctx, small) == 0
    rows, tiles = ctypes.c_int(), ctypes.c_int()
    assert lib.tapir_conv_plan(ctx, H, W, cin, cout, ks, stride, ctypes.byref(rows), ctypes.byref(tiles)) == 0
    y = np.zeros((N, Ho, Wo, cout), np.uint16)
    part = np.full((N, tiles.value, cout, 2), np.nan, np.float32)
    ss = np.zeros((N, cin, 2), np.float32)
    rc = lib.tapir_conv_fused(ctx, _p(xb), _p(part_in), 2, 0, _p(gamma), _p(beta), _p(ss), ws,
                              _p(to_bf16_bits(sc)) if shortcut else None, _p(y), _p(part), N, H, W, cin, cout, ks, stride, None)
    assert rc == 0, lib.tapir_last_error(ctx)
    return from_bf16_bits(y), part, rows.value, tiles.value

  try:
    big, _, _, _ = run(0)
    got, part, rows, tiles = run(1)
  finally:
    assert lib.tapir_conv_set_small(ctx, 0) == 0
  assert rows * Wo <= 128 and tiles == -(-Ho // rows)
  a, b = _norm_pairs(x, gamma, beta)
  ref = _conv_ref(_r(np.maximum(x * a + b, 0)), _r(w), stride)
  if shortcut:
    ref = ref + sc
  np.testing.assert_allclose(got, ref, atol=2e-2, rtol=1e-2)
  assert np.abs(got - ref).mean() < 2e-3
  # the two forms: the same values up to the summation order (one bf16 step on a few outputs)
  d = np.abs(got - big)
  assert d.max() <= 2e-2 * max(1.0, np.abs(big).max()) and (d > 0).mean() < 0.05, (d.max(), (d > 0).mean())
  # the tile summaries describe the STORED tensor
  cnt = np.array([min(rows, Ho - t * rows) * Wo for t in range(tiles)], np.float64)
  pm, pM2 = part[..., 0].astype(np.float64), part[..., 1].astype(np.float64)
  assert np.isfinite(part).all()
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  np.testing.assert_allclose(tot_mean, got.mean((1, 2), dtype=np.float64), atol=1e-5)
  np.testing.assert_allclose(tot_M2 / (Ho * Wo), got.astype(np.float64).var((1, 2)), rtol=1e-4, atol=1e-6)
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('C,H,W', [(256, 6, 32), (64, 5, 128), (128, 70, 12)])
def test_few_frame_form_merges_the_next_norm_in_the_launch(C, H, W):
  """tapir_conv_fused_nn in the few-frame form: an image's LAST arriver among its tiles x C / 64 workgroups merges the tile
  summaries into the next norm's (a, b) pairs -- the same pairs the consuming call's own merge launch computes from the
  summaries, bit for bit; counters back at zero; y and the summaries as without the merge.  (70 x 12: 14 tiles.)"""
  lib = emu_lib()
  ctx = _ctx(lib)
  rng = np.random.default_rng(C + H)
  N = 2
  x = _r(rng.standard_normal((N, H, W, C)) * 1.5 + 0.5)
  w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  g0, b0 = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
  g1, b1 = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
  xb = to_bf16_bits(x)
  part_in = np.zeros((N, 2, C, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, C, 2, None) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(np.ascontiguousarray(w)), C, C, 3, ctypes.byref(ws)) == 0
  assert lib.tapir_conv_set_small(ctx, 1) == 0
  try:
    rows, tiles = ctypes.c_int(), ctypes.c_int()
    assert lib.tapir_conv_plan(ctx, H, W, C, C, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0

    def call(nn):
      y = np.zeros((N, H, W, C), np.uint16)
      part = np.zeros((N, tiles.value, C, 2), np.float32)
      ss = np.zeros((N, C, 2), np.float32)
      rc = lib.tapir_conv_fused_nn(ctx, _p(xb), _p(part_in), 2, 0, _p(g0), _p(b0), _p(ss), ws, None, _p(y), _p(part),
                                   N, H, W, C, C, 3, 1, ctypes.byref(nn) if nn is not None else None, None)
      assert rc == 0, lib.tapir_last_error(ctx)
      return y, part

    y_ref, part_ref = call(None)
    # the consuming call's own merge LAUNCH (the many-frame path: in the few-frame form the consumer merges the summaries in
    # its prologue and writes no `ss`; both read the same summaries with the same arithmetic)
    ss_ref = np.zeros((N, C, 2), np.float32)
    y2 = np.zeros((N, H, W, C), np.uint16)
    assert lib.tapir_conv_set_small(ctx, 0) == 0
    assert lib.tapir_conv_fused(ctx, _p(y_ref), _p(part_ref), tiles.value, rows.value * W, _p(g1), _p(b1), _p(ss_ref), ws,
                                None, _p(y2), None, N, H, W, C, C, 3, 1, None) == 0
    assert lib.tapir_conv_set_small(ctx, 1) == 0
    # ... and the few-frame consumer on the same summaries: the same convolution up to the summation order
    y3 = np.zeros((N, H, W, C), np.uint16)
    ss_unused = np.full((N, C, 2), np.nan, np.float32)
    assert lib.tapir_conv_fused(ctx, _p(y_ref), _p(part_ref), tiles.value, rows.value * W, _p(g1), _p(b1), _p(ss_unused), ws,
                                None, _p(y3), None, N, H, W, C, C, 3, 1, None) == 0
    d23 = np.abs(from_bf16_bits(y3) - from_bf16_bits(y2))
    assert d23.max() <= 2e-2 * max(1.0, np.abs(from_bf16_bits(y2)).max()) and (d23 > 0).mean() < 0.05
    assert np.isnan(ss_unused).all()       # (merged in the kernel's prologue: `ss` is not written)
    ssn = np.full((N, C, 2), np.nan, np.float32)
    arrive = np.zeros(N, np.int32)
    nn = _ffi.TapirNextNorm(g1.ctypes.data, b1.ctypes.data, ssn.ctypes.data, arrive.ctypes.data)
    for _ in range(2):          # twice: the counters come back to zero
      y, part = call(nn)
      assert (arrive == 0).all()
      np.testing.assert_array_equal(y, y_ref)
      np.testing.assert_array_equal(part, part_ref)
      np.testing.assert_array_equal(ssn, ss_ref)
      ssn[:] = np.nan
  finally:
    assert lib.tapir_conv_set_small(ctx, 0) == 0
  lib.tapir_destroy(ctx)
