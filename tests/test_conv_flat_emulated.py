"""The flat tiling of the 3x3 256 -> 256 block convolutions (csrc/conv_flat.hpp: three consecutive 64-pixel slabs of the
launch's (image, row) space per workgroup, image boundaries wherever they fall) against the per-image tiling it
replaces (csrc/conv_fused.hpp), on the fiber emulator: outputs, slab summaries and the in-launch merged (a, b) pairs of
the next norm are BIT-identical -- with and without the shortcut, with the 1x1 projection fused in (DUAL), with image
boundaries inside workgroups, a ragged last workgroup, and slabs of fewer than 64 pixels.  The per-image kernel itself
is held to numpy by tests/test_conv_fused_emulated.py (resnet.py:241-256)."""
import ctypes

import numpy as np
import pytest

from tapnet_amd import _ffi
from tests.emu_engine import emu_lib
from tests.test_conv_fused_emulated import _ctx, _p, _r
from tests.test_gemm_tiles_emulated import to_bf16_bits

C = 256


def _both(lib, ctx, fn, N, H, W):
  outs = []
  for mode in (0, 1):
    assert lib.tapir_debug_set_conv_flat(ctx, mode) == 0
    wgs = ctypes.c_int()     # the launch below really takes the form under test
    rc = lib.tapir_conv_flat_plan(ctx, N, H, W, C, C, 3, 1, ctypes.byref(wgs))
    assert (rc == 0 and wgs.value == -(-N * (H // (64 // W)) // 3)) if mode else rc == _ffi.TAPIR_ERR_UNSUPPORTED
    outs.append(fn())
  assert lib.tapir_debug_set_conv_flat(ctx, 0) == 0
  return outs


@pytest.mark.parametrize('N,H,W,shortcut', [
    (3, 8, 32, True),     # 4 slabs per image, 12 slabs = 4 workgroups: two of them straddle an image boundary
    (2, 10, 32, False),   # 5 slabs per image, 10 slabs: ragged last workgroup (one slab), a boundary inside workgroup 1
    (2, 6, 24, True),     # slabs of 48 pixels (2 rows of 24): a masked fragment per slab
    (1, 9, 20, False),    # slabs of 3 rows x 20 = 60 pixels, one image, one workgroup
    (4, 6, 32, True),     # 3 slabs per image: every workgroup is exactly one image
])
def test_flat_tiling_is_bit_identical(N, H, W, shortcut):
  lib = emu_lib()
  ctx = _ctx(lib)
  rng = np.random.default_rng(N * 1000 + H * 10 + W)
  x = _r(rng.standard_normal((N, H, W, C)) * 1.5 + 0.5)
  w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  g0, b0 = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
  g1, b1 = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
  sc = to_bf16_bits(_r(rng.standard_normal((N, H, W, C)))) if shortcut else None
  xb = to_bf16_bits(x)
  part_in = np.zeros((N, 2, C, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, C, 2, None) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, _p(np.ascontiguousarray(w)), C, C, 3, ctypes.byref(ws)) == 0
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, H, W, C, C, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  assert H % rows.value == 0 and tiles.value >= 3

  def run(nn_on):
    y = np.zeros((N, H, W, C), np.uint16)
    part = np.full((N, tiles.value, C, 2), np.nan, np.float32)
    ss = np.zeros((N, C, 2), np.float32)
    ssn, arrive = np.full((N, C, 2), np.nan, np.float32), np.zeros(N, np.int32)
    nn = _ffi.TapirNextNorm(g1.ctypes.data, b1.ctypes.data, ssn.ctypes.data, arrive.ctypes.data)
    rc = lib.tapir_conv_fused_nn(ctx, _p(xb), _p(part_in), 2, 0, _p(g0), _p(b0), _p(ss), ws, _p(sc), _p(y), _p(part),
                                 N, H, W, C, C, 3, 1, ctypes.byref(nn) if nn_on else None, None)
    assert rc == 0, lib.tapir_last_error(ctx)
    assert (arrive == 0).all()
    return y, part, ssn

  for nn_on in (True, False):
    (y0, p0, s0), (y1, p1, s1) = _both(lib, ctx, lambda: run(nn_on), N, H, W)
    assert np.isfinite(p0).all()
    np.testing.assert_array_equal(y1, y0)
    np.testing.assert_array_equal(p1, p0)
    if nn_on:
      assert np.isfinite(s0).all()
      np.testing.assert_array_equal(s1, s0)
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('N,H,W', [(3, 8, 32), (2, 12, 16)])
def test_flat_tiling_with_the_projection_fused_in(N, H, W):
  """DUAL: conv_0 + proj_conv of a stride-1 256 -> 256 block (ResNet group 3's first block) in one launch, flat."""
  lib = emu_lib()
  ctx = _ctx(lib)
  rng = np.random.default_rng(N + H + W)
  x = _r(rng.standard_normal((N, H, W, C)) * 1.5 + 0.5)
  w3 = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  w1 = (rng.standard_normal((C, C, 1, 1)) / np.sqrt(C)).astype(np.float32)
  g0, b0 = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
  g1, b1 = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.standard_normal(C) * 0.3).astype(np.float32)
  xb = to_bf16_bits(x)
  part_in = np.zeros((N, 2, C, 2), np.float32)
  assert lib.tapir_inorm_stats(ctx, _p(xb), None, None, _p(part_in), N, H * W, C, 2, None) == 0
  wsd = ctypes.c_void_p()
  assert lib.tapir_conv_pack_dual(ctx, _p(w3), _p(w1), C, C, 1, ctypes.byref(wsd)) == 0
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, H, W, C, C, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  assert H % rows.value == 0 and tiles.value >= 3

  def run():
    y, yp = np.zeros((N, H, W, C), np.uint16), np.zeros((N, H, W, C), np.uint16)
    part = np.full((N, tiles.value, C, 2), np.nan, np.float32)
    ss = np.zeros((N, C, 2), np.float32)
    ssn, arrive = np.full((N, C, 2), np.nan, np.float32), np.zeros(N, np.int32)
    nn = _ffi.TapirNextNorm(g1.ctypes.data, b1.ctypes.data, ssn.ctypes.data, arrive.ctypes.data)
    rc = lib.tapir_conv_fused_dual_nn(ctx, _p(xb), _p(part_in), 2, 0, _p(g0), _p(b0), _p(ss), wsd, _p(y), _p(yp), _p(part),
                                      N, H, W, C, C, 1, ctypes.byref(nn), None)
    assert rc == 0, lib.tapir_last_error(ctx)
    assert (arrive == 0).all()
    return y, yp, part, ssn

  a, b = _both(lib, ctx, run, N, H, W)
  for u, v in zip(a, b):
    np.testing.assert_array_equal(v, u)
  assert np.isfinite(a[2]).all() and np.isfinite(a[3]).all()
  lib.tapir_destroy(ctx)
