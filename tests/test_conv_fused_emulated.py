"""The fused 3x3 backbone convolution (csrc/conv_fused.hpp: InstanceNorm + ReLU folded into the operand
load, residual add and next-norm statistics in the epilogue) on the fiber emulator, against a plain
numpy restatement of resnet.py:241-256 on the same bf16-rounded operands."""
import ctypes

import numpy as np
import pytest

from tapnet_amd import _ffi
from tests.emu_engine import emu_lib
from tests.test_gemm_tiles_emulated import from_bf16_bits, to_bf16_bits


def _ctx(lib, dtype=_ffi.TAPIR_BF16):
  cfg = _ffi.TapirCfg(1, 4, 1, 0, 20.0, 64, 64, dtype)
  ctx = ctypes.c_void_p()
  assert lib.tapir_create(ctypes.byref(ctx), ctypes.byref(cfg), 0) == 0
  return ctx


def _p(x):
  return None if x is None else x.ctypes.data_as(ctypes.c_void_p)


def _r(x):
  return from_bf16_bits(to_bf16_bits(np.asarray(x, np.float32)))


def _conv_ref(xn, w, stride=1):
  """xn [N,H,W,Ci] (already normalised), w [Co,Ci,k,k]; XLA SAME zero padding; float64 accumulation."""
  N, H, W, C = xn.shape
  k = w.shape[2]
  Ho, Wo = -(-H // stride), -(-W // stride)
  ty, tx = max((Ho - 1) * stride + k - H, 0), max((Wo - 1) * stride + k - W, 0)
  xp = np.zeros((N, H + ty, W + tx, C), np.float64)
  xp[:, ty // 2:ty // 2 + H, tx // 2:tx // 2 + W] = xn
  out = np.zeros((N, Ho, Wo, w.shape[0]), np.float64)
  for ky in range(k):
    for kx in range(k):
      out += xp[:, ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride] @ \
          w[:, :, ky, kx].astype(np.float64).T
  return out


def run_conv(lib, ctx, x_bits, part_in, slabs_in, per_s_in, gamma, beta, wstream, shortcut_bits, N, H, W, C,
             cout=None, ks=3, stride=1):
  cout = cout or C
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, H, W, C, cout, ks, stride, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  Ho, Wo = -(-H // stride), -(-W // stride)
  y = np.zeros((N, Ho, Wo, cout), np.uint16)
  part = np.zeros((N, tiles.value, cout, 2), np.float32)
  ss = np.zeros((N, C, 2), np.float32)
  rc = lib.tapir_conv_fused(ctx, _p(x_bits), _p(part_in), slabs_in, per_s_in, _p(gamma), _p(beta), _p(ss),
                            wstream, _p(shortcut_bits), _p(y), _p(part), N, H, W, C, cout, ks, stride, None)
  assert rc == 0, lib.tapir_last_error(ctx)
  return y, part, rows.value, tiles.value


@pytest.mark.parametrize('C,H,W,shortcut', [(64, 8, 16, True), (64, 5, 24, False), (128, 6, 12, True),
                                            (256, 9, 8, True), (256, 4, 32, False),
                                            # rows too long for the 4-wave tile: the 8-wave workgroups
                                            (64, 3, 200, True), (128, 3, 100, False), (256, 4, 40, True)])
def test_conv3x3_fused(C, H, W, shortcut):
  lib = emu_lib()
  ctx = _ctx(lib)
  rng = np.random.default_rng(C + H + W)
  N = 2
  x = _r(rng.standard_normal((N, H, W, C)) * 1.5 + 0.5)
  w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, C).astype(np.float32)
  beta = (rng.standard_normal(C) * 0.3).astype(np.float32)
  sc = _r(rng.standard_normal((N, H, W, C))) if shortcut else None
  xb = to_bf16_bits(x)
  slabs = 3
  part_in = np.zeros((N, slabs, C, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, C, slabs, None) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(np.ascontiguousarray(w)), C, C, 3, ctypes.byref(ws)) == 0
  y, part, rows, tiles = run_conv(lib, ctx, xb, part_in, slabs, 0, gamma, beta, ws,
                                  to_bf16_bits(sc) if shortcut else None, N, H, W, C)
  # reference on the same rounded operands
  mean = x.mean((1, 2), keepdims=True, dtype=np.float64)
  var = x.astype(np.float64).var((1, 2), keepdims=True)
  rstd = 1.0 / np.sqrt(var + 1e-5)
  a = (rstd * gamma).astype(np.float32)
  b = (beta - mean * rstd * gamma).astype(np.float32)
  xn = _r(np.maximum(x * a + b, 0))
  ref = _conv_ref(xn, _r(w))
  if shortcut:
    ref = ref + sc
  got = from_bf16_bits(y)
  # bf16 output rounding (2^-9 relative) + the (a, b) pair computed in f32 from merged summaries
  np.testing.assert_allclose(got, ref, atol=2e-2, rtol=1e-2)
  assert np.abs(got - ref).mean() < 2e-3
  # the summaries describe the STORED tensor exactly: merge them and compare with numpy
  assert part.shape[1] == tiles and tiles == -(-H // rows)
  cnt = np.array([min(rows, H - t * rows) * W for t in range(tiles)], np.float64)
  pm, pM2 = part[..., 0].astype(np.float64), part[..., 1].astype(np.float64)
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  np.testing.assert_allclose(tot_mean, got.mean((1, 2), dtype=np.float64), atol=1e-5)
  np.testing.assert_allclose(tot_M2 / (H * W), got.astype(np.float64).var((1, 2)), rtol=1e-4, atol=1e-6)
  lib.tapir_destroy(ctx)


def test_conv3x3_chain_uses_part_out():
  """part_out of one call is a valid part_in of the next (slabs = tiles, per_s = rows * W)."""
  lib = emu_lib()
  ctx = _ctx(lib)
  rng = np.random.default_rng(3)
  N, H, W, C = 1, 7, 16, 64
  x = _r(rng.standard_normal((N, H, W, C)))
  w0 = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  w1 = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  g = np.ones(C, np.float32)
  z = np.zeros(C, np.float32)
  xb = to_bf16_bits(x)
  part_in = np.zeros((N, 1, C, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, C, 1, None) == 0
  ws0, ws1 = ctypes.c_void_p(), ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(w0), C, C, 3, ctypes.byref(ws0)) == 0
  assert lib.tapir_conv_pack(ctx, _p(w1), C, C, 3, ctypes.byref(ws1)) == 0
  y0, part0, rows, tiles = run_conv(lib, ctx, xb, part_in, 1, 0, g, z, ws0, None, N, H, W, C)
  y1, _, _, _ = run_conv(lib, ctx, y0, part0, tiles, rows * W, g, z, ws1, xb, N, H, W, C)
  # the same second convolution with summaries recomputed by the stand-alone statistics kernel
  part_chk = np.zeros((N, 2, C, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(y0), None, None, _p(part_chk), N, H * W, C, 2, None) == 0
  y1b, _, _, _ = run_conv(lib, ctx, y0, part_chk, 2, 0, g, z, ws1, xb, N, H, W, C)
  d = np.abs(from_bf16_bits(y1) - from_bf16_bits(y1b))
  assert d.max() <= 2e-2 and (d > 0).mean() < 0.02   # (a, b) differ in the last f32 bits at most
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('cin,cout,ks,stride,H,W', [
    (64, 128, 3, 2, 8, 16), (64, 128, 3, 2, 7, 10), (128, 256, 3, 2, 6, 8), (64, 128, 1, 2, 8, 12),
    (128, 256, 1, 2, 5, 6), (64, 64, 1, 1, 6, 20), (256, 256, 1, 1, 5, 8),
    (64, 128, 3, 2, 4, 140)])      # rows too long for the 4-wave tile
def test_strided_and_projection_convs(cin, cout, ks, stride, H, W):
  """conv_0 / proj_conv of the first block of a group (resnet.py:243-247): 3x3 and 1x1, stride 1 and 2,
  XLA SAME padding (one row / column on the HIGH side for even sizes at stride 2, one on each side for
  odd sizes)."""
  lib = emu_lib()
  ctx = _ctx(lib)
  rng = np.random.default_rng(cin + cout + ks + H + W)
  N = 2
  x = _r(rng.standard_normal((N, H, W, cin)) * 1.5 + 0.5)
  w = (rng.standard_normal((cout, cin, ks, ks)) / np.sqrt(ks * ks * cin)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, cin).astype(np.float32)
  beta = (rng.standard_normal(cin) * 0.3).astype(np.float32)
  xb = to_bf16_bits(x)
  part_in = np.zeros((N, 2, cin, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, cin, 2, None) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(np.ascontiguousarray(w)), cout, cin, ks, ctypes.byref(ws)) == 0
  y, part, rows, tiles = run_conv(lib, ctx, xb, part_in, 2, 0, gamma, beta, ws, None, N, H, W, cin, cout, ks, stride)
  mean = x.mean((1, 2), keepdims=True, dtype=np.float64)
  var = x.astype(np.float64).var((1, 2), keepdims=True)
  rstd = 1.0 / np.sqrt(var + 1e-5)
  xn = _r(np.maximum(x * (rstd * gamma).astype(np.float32) + (beta - mean * rstd * gamma).astype(np.float32), 0))
  ref = _conv_ref(xn, _r(w), stride)
  got = from_bf16_bits(y)
  assert got.shape == ref.shape
  np.testing.assert_allclose(got, ref, atol=2e-2, rtol=1e-2)
  assert np.abs(got - ref).mean() < 2e-3
  Ho, Wo = got.shape[1:3]
  cnt = np.array([min(rows, Ho - t * rows) * Wo for t in range(tiles)], np.float64)
  pm, pM2 = part[..., 0].astype(np.float64), part[..., 1].astype(np.float64)
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  np.testing.assert_allclose(tot_mean, got.mean((1, 2), dtype=np.float64), atol=1e-5)
  np.testing.assert_allclose(tot_M2 / (Ho * Wo), got.astype(np.float64).var((1, 2)), rtol=1e-4, atol=1e-6)
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('dtype', [_ffi.TAPIR_BF16, _ffi.TAPIR_F32])
@pytest.mark.parametrize('H,W', [(16, 32), (9, 12), (6, 300)])
def test_stem_conv(H, W, dtype):
  """7x7 / stride 2 / SAME stem (resnet.py:356-364) on f32 frames, against numpy (on the bf16-rounded
  operands for the bf16 build; the f32 build at 1e-4); summaries of the stored output."""
  lib = emu_lib()
  ctx = _ctx(lib, dtype)
  bf = dtype == _ffi.TAPIR_BF16
  rng = np.random.default_rng(H + W)
  N = 2
  x = rng.uniform(-1, 1, (N, H, W, 3)).astype(np.float32)
  w = (rng.standard_normal((64, 3, 7, 7)) / np.sqrt(147)).astype(np.float32)
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_stem_plan(ctx, H, W, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_stem_pack(ctx, _p(w), ctypes.byref(ws)) == 0
  Ho, Wo = -(-H // 2), -(-W // 2)
  y = np.zeros((N, Ho, Wo, 64), np.uint16 if bf else np.float32)
  part = np.zeros((N, tiles.value, 64, 2), np.float32)
  rc = lib.tapir_stem_conv(ctx, _p(x), ws, _p(y), _p(part), N, H, W, None)
  assert rc == 0, lib.tapir_last_error(ctx)
  if bf:
    ref = _conv_ref(_r(x), _r(w), 2)
    got = from_bf16_bits(y)
    np.testing.assert_allclose(got, ref, atol=1e-2, rtol=1e-2)
    assert np.abs(got - ref).mean() < 1e-3
  else:
    ref = _conv_ref(x, w, 2)
    got = y
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=1e-4)
  cnt = np.array([min(rows.value, Ho - t * rows.value) * Wo for t in range(tiles.value)], np.float64)
  pm, pM2 = part[..., 0].astype(np.float64), part[..., 1].astype(np.float64)
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  np.testing.assert_allclose(tot_mean, got.mean((1, 2), dtype=np.float64), atol=1e-5)
  np.testing.assert_allclose(tot_M2 / (Ho * Wo), got.astype(np.float64).var((1, 2)), rtol=1e-4, atol=1e-6)
  # tapir_stem_conv_nn: the first InstanceNorm's (a, b) pairs merged inside the launch == the merge launch of the
  # convolution that consumes the stem's output
  g1, b1 = rng.uniform(0.5, 1.5, 64).astype(np.float32), (rng.standard_normal(64) * 0.3).astype(np.float32)
  wc = (rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)
  wcs = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(np.ascontiguousarray(wc)), 64, 64, 3, ctypes.byref(wcs)) == 0
  ss_ref = np.zeros((N, 64, 2), np.float32)
  y2 = np.zeros((N, Ho, Wo, 64), y.dtype)
  assert lib.tapir_conv_fused(ctx, _p(y), _p(part), tiles.value, rows.value * Wo, _p(g1), _p(b1), _p(ss_ref), wcs, None,
                              _p(y2), None, N, Ho, Wo, 64, 64, 3, 1, None) == 0
  ssn, arrive = np.full((N, 64, 2), np.nan, np.float32), np.zeros(N, np.int32)
  nn = _ffi.TapirNextNorm(g1.ctypes.data, b1.ctypes.data, ssn.ctypes.data, arrive.ctypes.data)
  y3, part3 = np.zeros_like(y), np.zeros_like(part)
  assert lib.tapir_stem_conv_nn(ctx, _p(x), ws, _p(y3), _p(part3), N, H, W, ctypes.byref(nn), None) == 0
  np.testing.assert_array_equal(y3, y)
  np.testing.assert_array_equal(part3, part)
  np.testing.assert_array_equal(ssn, ss_ref)
  assert (arrive == 0).all()
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('cin,cout,ks,stride,H,W,shortcut', [
    (64, 64, 3, 1, 6, 16, True), (128, 128, 3, 1, 5, 12, False), (256, 256, 3, 1, 4, 8, True),
    (64, 128, 3, 2, 7, 10, False), (128, 256, 1, 2, 6, 8, False), (64, 64, 1, 1, 4, 20, False),
    (64, 64, 3, 1, 3, 130, True)])      # the 8-wave tile
def test_conv_fused_f32(cin, cout, ks, stride, H, W, shortcut):
  """The f32 instantiation (exact-f32 MFMA, the parity build): the same kernel structure against float64
  numpy at 1e-4 -- no operand rounding anywhere."""
  lib = emu_lib()
  ctx = _ctx(lib, _ffi.TAPIR_F32)
  rng = np.random.default_rng(cin + cout + ks + H)
  N = 2
  x = (rng.standard_normal((N, H, W, cin)) * 1.5 + 0.5).astype(np.float32)
  w = (rng.standard_normal((cout, cin, ks, ks)) / np.sqrt(ks * ks * cin)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, cin).astype(np.float32)
  beta = (rng.standard_normal(cin) * 0.3).astype(np.float32)
  Ho, Wo = -(-H // stride), -(-W // stride)
  sc = rng.standard_normal((N, Ho, Wo, cout)).astype(np.float32) if shortcut else None
  part_in = np.zeros((N, 2, cin, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(x), None, None, _p(part_in), N, H * W, cin, 2, None) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(np.ascontiguousarray(w)), cout, cin, ks, ctypes.byref(ws)) == 0
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, H, W, cin, cout, ks, stride, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  y = np.zeros((N, Ho, Wo, cout), np.float32)
  part = np.zeros((N, tiles.value, cout, 2), np.float32)
  ss = np.zeros((N, cin, 2), np.float32)
  rc = lib.tapir_conv_fused(ctx, _p(x), _p(part_in), 2, 0, _p(gamma), _p(beta), _p(ss), ws, _p(sc), _p(y), _p(part),
                            N, H, W, cin, cout, ks, stride, None)
  assert rc == 0, lib.tapir_last_error(ctx)
  mean = x.mean((1, 2), keepdims=True, dtype=np.float64)
  var = x.astype(np.float64).var((1, 2), keepdims=True)
  xn = np.maximum((x - mean) / np.sqrt(var + 1e-5) * gamma + beta, 0)
  ref = _conv_ref(xn, w, stride)
  if shortcut:
    ref = ref + sc
  np.testing.assert_allclose(y, ref, atol=1e-4, rtol=1e-4)
  cnt = np.array([min(rows.value, Ho - t * rows.value) * Wo for t in range(tiles.value)], np.float64)
  pm, pM2 = part[..., 0].astype(np.float64), part[..., 1].astype(np.float64)
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  np.testing.assert_allclose(tot_mean, y.mean((1, 2), dtype=np.float64), atol=1e-5)
  np.testing.assert_allclose(tot_M2 / (Ho * Wo), y.astype(np.float64).var((1, 2)), rtol=1e-4, atol=1e-6)
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('cin,cout,stride,H,W,dtype', [
    (64, 64, 1, 6, 20, _ffi.TAPIR_BF16), (256, 256, 1, 5, 8, _ffi.TAPIR_BF16), (64, 128, 2, 8, 16, _ffi.TAPIR_BF16),
    (64, 128, 2, 7, 10, _ffi.TAPIR_BF16), (128, 256, 2, 6, 9, _ffi.TAPIR_BF16), (64, 128, 2, 4, 140, _ffi.TAPIR_BF16),
    (64, 64, 1, 5, 12, _ffi.TAPIR_F32), (128, 256, 2, 5, 6, _ffi.TAPIR_F32)])
def test_conv0_and_projection_in_one_launch(cin, cout, stride, H, W, dtype):
  """tapir_conv_fused_dual_nn: conv_0 (3x3) and proj_conv (1x1, same stride) of a group's first block
  (resnet.py:232-247: both read relu(bn_0(x))) from ONE staging of the input -- the projection is the centre tap
  (odd sizes at stride 2: tap (1, 1); even: tap (0, 0); mixed parities: (7, 10)).  Both outputs, the tile summaries
  and the in-launch merged pairs of the next norm are BIT-identical to the two separate launches."""
  lib = emu_lib()
  ctx = _ctx(lib, dtype)
  bf = dtype == _ffi.TAPIR_BF16
  rng = np.random.default_rng(cin + cout + H + W)
  N = 2
  x = _r(rng.standard_normal((N, H, W, cin)) * 1.5 + 0.5)
  w3 = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
  w1 = (rng.standard_normal((cout, cin, 1, 1)) / np.sqrt(cin)).astype(np.float32)
  g0, b0 = rng.uniform(0.5, 1.5, cin).astype(np.float32), (rng.standard_normal(cin) * 0.3).astype(np.float32)
  g1, b1 = rng.uniform(0.5, 1.5, cout).astype(np.float32), (rng.standard_normal(cout) * 0.3).astype(np.float32)
  xb = to_bf16_bits(x) if bf else x.astype(np.float32)
  part_in = np.zeros((N, 2, cin, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, cin, 2, None) == 0
  ws3, ws1, wsd = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(w3), cout, cin, 3, ctypes.byref(ws3)) == 0
  assert lib.tapir_conv_pack(ctx, _p(w1), cout, cin, 1, ctypes.byref(ws1)) == 0
  assert lib.tapir_conv_pack_dual(ctx, _p(w3), _p(w1), cout, cin, stride, ctypes.byref(wsd)) == 0
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, H, W, cin, cout, 3, stride, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  Ho, Wo = -(-H // stride), -(-W // stride)
  ydt = np.uint16 if bf else np.float32
  # the two separate launches (proj_conv merges bn_0's pairs, conv_0 re-uses them)
  ss = np.zeros((N, cin, 2), np.float32)
  yp_ref, y_ref = np.zeros((N, Ho, Wo, cout), ydt), np.zeros((N, Ho, Wo, cout), ydt)
  part_ref = np.zeros((N, tiles.value, cout, 2), np.float32)
  ssn_ref, arrive = np.full((N, cout, 2), np.nan, np.float32), np.zeros(N, np.int32)
  nn = _ffi.TapirNextNorm(g1.ctypes.data, b1.ctypes.data, ssn_ref.ctypes.data, arrive.ctypes.data)
  assert lib.tapir_conv_fused_nn(ctx, _p(xb), _p(part_in), 2, 0, _p(g0), _p(b0), _p(ss), ws1, None, _p(yp_ref), None,
                                 N, H, W, cin, cout, 1, stride, None, None) == 0, lib.tapir_last_error(ctx)
  assert lib.tapir_conv_fused_nn(ctx, _p(xb), None, 2, 0, _p(g0), _p(b0), _p(ss), ws3, None, _p(y_ref), _p(part_ref),
                                 N, H, W, cin, cout, 3, stride, ctypes.byref(nn), None) == 0, lib.tapir_last_error(ctx)
  # one launch
  ss2 = np.zeros((N, cin, 2), np.float32)
  yp, y = np.zeros_like(yp_ref), np.zeros_like(y_ref)
  part = np.zeros_like(part_ref)
  ssn = np.full((N, cout, 2), np.nan, np.float32)
  nn2 = _ffi.TapirNextNorm(g1.ctypes.data, b1.ctypes.data, ssn.ctypes.data, arrive.ctypes.data)
  rc = lib.tapir_conv_fused_dual_nn(ctx, _p(xb), _p(part_in), 2, 0, _p(g0), _p(b0), _p(ss2), wsd, _p(y), _p(yp), _p(part),
                                    N, H, W, cin, cout, stride, ctypes.byref(nn2), None)
  assert rc == 0, lib.tapir_last_error(ctx)
  np.testing.assert_array_equal(ss2, ss)
  np.testing.assert_array_equal(y, y_ref)
  np.testing.assert_array_equal(yp, yp_ref)
  np.testing.assert_array_equal(part, part_ref)
  np.testing.assert_array_equal(ssn, ssn_ref)
  assert (arrive == 0).all() and np.abs(from_bf16_bits(yp) if bf else yp).max() > 0.1
  # shapes the dual form does not take
  assert lib.tapir_conv_pack_dual(ctx, _p(w3), _p(w1), cout, cin, 3 - stride, ctypes.byref(wsd)) == _ffi.TAPIR_ERR_UNSUPPORTED
  lib.tapir_destroy(ctx)


def test_conv_rejects_bad_shapes():
  lib = emu_lib()
  ctx = _ctx(lib, _ffi.TAPIR_F32)
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  plan = lambda c_, h, w, ci, co, k, s: lib.tapir_conv_plan(c_, h, w, ci, co, k, s, ctypes.byref(rows), ctypes.byref(tiles))
  assert plan(ctx, 8, 8, 64, 64, 3, 1) == 0                                   # f32 contexts: the same kernels, exact-f32 MFMA
  assert plan(ctx, 128, 128, 64, 64, 3, 1) == 0 and rows.value >= 1
  assert lib.tapir_stem_plan(ctx, 64, 64, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  assert lib.tapir_stem_plan(ctx, 64, 63, ctypes.byref(rows), ctypes.byref(tiles)) == _ffi.TAPIR_ERR_UNSUPPORTED   # odd width
  lib.tapir_destroy(ctx)
  ctx = _ctx(lib)
  assert plan(ctx, 8, 8, 96, 96, 3, 1) == _ffi.TAPIR_ERR_UNSUPPORTED
  assert plan(ctx, 8, 8, 64, 256, 3, 2) == _ffi.TAPIR_ERR_UNSUPPORTED      # not a ResNet block shape
  assert plan(ctx, 8, 8, 64, 64, 5, 1) == _ffi.TAPIR_ERR_UNSUPPORTED
  assert plan(ctx, 8, 2000, 256, 256, 3, 1) == _ffi.TAPIR_ERR_UNSUPPORTED
  assert plan(ctx, 128, 128, 64, 64, 3, 1) == 0
  assert (rows.value, tiles.value) == (2, 64)      # 4-wave workgroups: 256 pixels
  assert plan(ctx, 256, 256, 64, 64, 3, 1) == 0
  assert (rows.value, tiles.value) == (2, 128)     # rows too long for the 72-KiB tile: 8 waves, 512 pixels
  assert plan(ctx, 128, 128, 64, 128, 3, 2) == 0 and rows.value >= 2
  assert plan(ctx, 128, 128, 64, 128, 1, 2) == 0
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('C,H,W,dtype', [(64, 40, 24, _ffi.TAPIR_BF16), (256, 9, 8, _ffi.TAPIR_BF16),
                                         (128, 70, 12, _ffi.TAPIR_BF16), (64, 12, 16, _ffi.TAPIR_F32)])
def test_next_norm_pairs_merged_in_the_launch(C, H, W, dtype):
  """tapir_conv_fused_nn: the workgroup that draws the last ticket of an image merges the tile summaries into the
  (a, b) pairs of the NEXT InstanceNorm -- bit-identical to the separate merge launch of the consuming call, counters
  back at zero, y and the tile summaries unchanged.  (70 x 12: more tiles than one round of the merge holds.)"""
  lib = emu_lib()
  ctx = _ctx(lib, dtype)
  bf = dtype == _ffi.TAPIR_BF16
  rng = np.random.default_rng(C + H)
  N = 3
  x = _r(rng.standard_normal((N, H, W, C)) * 1.5 + 0.5)
  w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  g0, b0 = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
  g1, b1 = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
  xb = to_bf16_bits(x) if bf else x.astype(np.float32)
  slabs = 2
  part_in = np.zeros((N, slabs, C, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, C, slabs, None) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(np.ascontiguousarray(w)), C, C, 3, ctypes.byref(ws)) == 0
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, H, W, C, C, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  ydt = np.uint16 if bf else np.float32

  def call(nn):
    y = np.zeros((N, H, W, C), ydt)
    part = np.zeros((N, tiles.value, C, 2), np.float32)
    ss = np.zeros((N, C, 2), np.float32)
    rc = lib.tapir_conv_fused_nn(ctx, _p(xb), _p(part_in), slabs, 0, _p(g0), _p(b0), _p(ss), ws, None, _p(y), _p(part),
                                 N, H, W, C, C, 3, 1, ctypes.byref(nn) if nn is not None else None, None)
    assert rc == 0, lib.tapir_last_error(ctx)
    return y, part

  y_ref, part_ref = call(None)
  # the consuming call's own merge launch: its `ss` is what the _nn form has to reproduce
  ss_ref = np.zeros((N, C, 2), np.float32)
  y2 = np.zeros((N, H, W, C), ydt)
  assert lib.tapir_conv_fused(ctx, _p(y_ref), _p(part_ref), tiles.value, rows.value * W, _p(g1), _p(b1), _p(ss_ref), ws,
                              None, _p(y2), None, N, H, W, C, C, 3, 1, None) == 0
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
  # and the consumer takes the pairs as they are (part_in = NULL)
  call(nn)
  y3 = np.zeros((N, H, W, C), ydt)
  assert lib.tapir_conv_fused(ctx, _p(y_ref), None, 0, 0, None, None, _p(ssn), ws, None, _p(y3), None, N, H, W, C, C, 3,
                              1, None) == 0
  np.testing.assert_array_equal(y3, y2)
  bad = _ffi.TapirNextNorm(g1.ctypes.data, b1.ctypes.data, ssn.ctypes.data, None)
  assert lib.tapir_conv_fused_nn(ctx, _p(xb), _p(part_in), slabs, 0, _p(g0), _p(b0), _p(ss_ref), ws, None, _p(y3), _p(part),
                                 N, H, W, C, C, 3, 1, ctypes.byref(bad), None) == _ffi.TAPIR_ERR_INVALID
  lib.tapir_destroy(ctx)
