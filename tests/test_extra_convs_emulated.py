"""BootsTAPIR's ExtraConvs kernels (csrc/extra_convs.hpp: per-pixel LayerNorm, 3x3 implicit-GEMM convolution
with bias + GELU / bias + skip in the epilogue, output-channel passes, input-channel chunks) on the fiber
emulator, against numpy restatements of tapnet/models/tapir_model.py:159-186 on the same rounded operands
(bf16 build) and in float64 (f32 build)."""
import ctypes

import numpy as np
import pytest

from tapnet_amd import _ffi
from tests.emu_engine import emu_lib
from tests.test_conv_fused_emulated import _conv_ref, _ctx, _p, _r
from tests.test_gemm_tiles_emulated import from_bf16_bits, to_bf16_bits


def _gelu(x):
  return 0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3)))


def _ln(x, g, b):
  x = x.astype(np.float64)
  m = x.mean(-1, keepdims=True)
  v = x.var(-1, keepdims=True)
  return (x - m) / np.sqrt(v + 1e-5) * g + b


@pytest.mark.parametrize('dtype', ['bf16', 'f32'])
def test_layernorm_affine(dtype):
  lib = emu_lib()
  bf = dtype == 'bf16'
  ctx = _ctx(lib, _ffi.TAPIR_BF16 if bf else _ffi.TAPIR_F32)
  rng = np.random.default_rng(0)
  P, C = 37, 256
  x = (rng.standard_normal((P, C)) * 2 + 0.7).astype(np.float32)
  g = rng.uniform(0.5, 1.5, C).astype(np.float32)
  b = (rng.standard_normal(C) * 0.2).astype(np.float32)
  if bf:
    x = _r(x)
    xin, y = to_bf16_bits(x), np.zeros((P, C), np.uint16)
  else:
    xin, y = x, np.zeros((P, C), np.float32)
  assert lib.tapir_layernorm_affine(ctx, _p(xin), _p(g), _p(b), _p(y), P, C, None) == 0
  got = from_bf16_bits(y) if bf else y
  np.testing.assert_allclose(got, _ln(x, g, b), atol=2e-2 if bf else 2e-5, rtol=1e-2 if bf else 0)
  lib.tapir_destroy(ctx)


def _xconv(lib, ctx, bf, x, w, bias, skip, gelu):
  N, H, W, cin = x.shape
  cout = w.shape[0]
  rows, tiles, cch = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_xconv_plan(ctx, H, W, cin, cout, ctypes.byref(rows), ctypes.byref(tiles), ctypes.byref(cch)) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_xconv_pack(ctx, _p(np.ascontiguousarray(w, np.float32)), cout, cin, cch.value, ctypes.byref(ws)) == 0
  if bf:
    xin, sk, y = to_bf16_bits(x), (to_bf16_bits(skip) if skip is not None else None), np.zeros((N, H, W, cout), np.uint16)
  else:
    xin, sk, y = x, skip, np.zeros((N, H, W, cout), np.float32)
  rc = lib.tapir_xconv(ctx, _p(xin), ws, _p(bias), _p(sk), _p(y), N, H, W, cin, cout, 1 if gelu else 0, None)
  assert rc == 0, lib.tapir_last_error(ctx)
  assert lib.tapir_conv_free(ctx, ws) == 0
  return (from_bf16_bits(y) if bf else y), rows.value, tiles.value, cch.value


# (H, W): 2 rows x 32 (the 256x256 model), ragged last tile, one row of 64 (512x512 inputs), 8x8 (one tile),
# a narrow map with unused pixel columns
# nt: pixel tiles per wave -- 0 = xconv_plan chooses (the wide form, 128 pixels per workgroup, wherever it gives a tile more
# rows), 4 / 8 = that form only (TAPIR_XCONV_NT)
@pytest.mark.parametrize('dtype,H,W,nt', [('bf16', 4, 32, 4), ('bf16', 5, 32, 4), ('bf16', 2, 64, 4), ('bf16', 8, 8, 4),
                                          ('bf16', 3, 24, 4), ('f32', 3, 32, 4), ('f32', 2, 64, 4), ('f32', 4, 8, 4),
                                          ('bf16', 4, 32, 0), ('bf16', 9, 32, 0), ('bf16', 3, 64, 0), ('bf16', 6, 6, 8),
                                          ('bf16', 7, 24, 0), ('bf16', 2, 80, 0), ('f32', 5, 32, 0), ('f32', 2, 64, 8),
                                          ('f32', 9, 12, 0)])
def test_xconv_up_gelu_and_down_skip(dtype, H, W, nt, monkeypatch):
  """conv 256 -> 1024 (+ bias, GELU: 4 output-channel passes) and conv 1024 -> 256 (+ bias + skip: input
  channels in 4-16 LDS chunks), asymmetric random operands; both forms of the kernel (64 / 128 pixels per workgroup):
  ragged last tiles, tiles of one row, unused pixel columns, maps wider than 64 cells (wide form only)."""
  lib = emu_lib()
  bf = dtype == 'bf16'
  monkeypatch.setenv('TAPIR_XCONV_NT', str(nt))        # read by tapir_create
  ctx = _ctx(lib, _ffi.TAPIR_BF16 if bf else _ffi.TAPIR_F32)
  rng = np.random.default_rng(H * 100 + W)
  N, C = 2, 256
  rd = _r if bf else (lambda a: np.asarray(a, np.float32))
  x = rd(rng.standard_normal((N, H, W, C)))
  w1 = (rng.standard_normal((4 * C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  b1 = (rng.standard_normal(4 * C) * 0.1).astype(np.float32)
  got, rows, tiles, cch = _xconv(lib, ctx, bf, x, w1, b1, None, True)
  r4, r8 = min(H, 64 // W), min(H, 128 // W)
  assert rows == (r4 if nt == 4 else r8 if nt == 8 else (r8 if r8 > r4 else r4)) and tiles == -(-H // rows)
  ref = _gelu(_conv_ref(x, rd(w1)) + b1)
  if bf:
    np.testing.assert_allclose(got, ref, atol=1.5e-2, rtol=1e-2)
    assert np.abs(got - ref).mean() < 2e-3
  else:
    np.testing.assert_allclose(got, ref, atol=2e-5)
  # second convolution of the block on the (rounded) hidden tensor, skip = the block's normalised input
  hdn = rd(ref)
  w2 = (rng.standard_normal((C, 4 * C, 3, 3)) / np.sqrt(9 * 4 * C)).astype(np.float32)
  b2 = (rng.standard_normal(C) * 0.1).astype(np.float32)
  got2, _, _, cch2 = _xconv(lib, ctx, bf, hdn, w2, b2, x, False)
  assert 1024 % cch2 == 0 and cch2 <= cch
  ref2 = _conv_ref(hdn, rd(w2)) + b2 + x
  if bf:
    np.testing.assert_allclose(got2, ref2, atol=2e-2, rtol=1e-2)
    assert np.abs(got2 - ref2).mean() < 3e-3
  else:
    np.testing.assert_allclose(got2, ref2, atol=5e-5)
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('H,W', [(4, 32), (5, 16), (2, 64), (3, 40)])
def test_xconv_up_gelu_in_the_few_frame_form(H, W, monkeypatch):
  """tapir_conv_set_small (clips of fewer than 4 frames: the online model): the 256 -> 1024 convolution with bias + GELU
  in the form of csrc/conv_small.hpp (a workgroup per (row tile, 16 output channels), the waves split the taps) against
  numpy and against xconv_kernel on the same operands (same values up to the summation order).  tapir_model.py:183-184."""
  lib = emu_lib()
  monkeypatch.setenv('TAPIR_XCONV_NT', '4')        # the form short clips take (tapir_xconv_plan_frames)
  ctx = _ctx(lib, _ffi.TAPIR_BF16)
  rng = np.random.default_rng(H * 100 + W + 1)
  N, C = 2, 256
  x = _r(rng.standard_normal((N, H, W, C)))
  w1 = (rng.standard_normal((4 * C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  b1 = (rng.standard_normal(4 * C) * 0.1).astype(np.float32)
  big, _, _, cch = _xconv(lib, ctx, True, x, w1, b1, None, True)
  assert cch == (256 if W <= 32 or W == 40 else 128)   # (64-wide maps: the 64-pixel tile holds 128-channel chunks only)
  assert lib.tapir_conv_set_small(ctx, 1) == 0
  try:
    got, _, _, _ = _xconv(lib, ctx, True, x, w1, b1, None, True)
  finally:
    assert lib.tapir_conv_set_small(ctx, 0) == 0
  ref = _gelu(_conv_ref(x, _r(w1)) + b1)
  np.testing.assert_allclose(got, ref, atol=1.5e-2, rtol=1e-2)
  assert np.abs(got - ref).mean() < 2e-3
  d = np.abs(got - big)
  if cch == 256:
    assert d.max() <= 2e-2 and 0 < (d > 0).mean() < 0.05, (d.max(), (d > 0).mean())   # (> 0: the other kernel really ran)
  else:
    assert d.max() == 0      # packs for narrower chunks stay with xconv_kernel
  # the block's second convolution (1024 -> 256, + bias + skip: four input-channel chunks) in the same form
  hdn = _r(ref)
  w2 = (rng.standard_normal((C, 4 * C, 3, 3)) / np.sqrt(9 * 4 * C)).astype(np.float32)
  b2 = (rng.standard_normal(C) * 0.1).astype(np.float32)
  big2, _, _, cch2 = _xconv(lib, ctx, True, hdn, w2, b2, x, False)
  ref2 = _conv_ref(hdn, _r(w2)) + b2 + x
  if cch2 == 256:     # (maps whose 64-pixel tile holds 256-channel chunks: the few-frame form applies)
    assert lib.tapir_conv_set_small(ctx, 1) == 0
    try:
      got2, _, _, _ = _xconv(lib, ctx, True, hdn, w2, b2, x, False)
    finally:
      assert lib.tapir_conv_set_small(ctx, 0) == 0
    np.testing.assert_allclose(got2, ref2, atol=2e-2, rtol=1e-2)
    assert np.abs(got2 - ref2).mean() < 3e-3
    d2 = np.abs(got2 - big2)
    assert d2.max() <= 4e-2 and 0 < (d2 > 0).mean() < 0.08, (d2.max(), (d2 > 0).mean())
  lib.tapir_destroy(ctx)


def test_xconv_rejects_unsupported_shapes():
  lib = emu_lib()
  ctx = _ctx(lib)
  r, t, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_xconv_plan(ctx, 8, 136, 256, 1024, ctypes.byref(r), ctypes.byref(t), ctypes.byref(c)) == _ffi.TAPIR_ERR_UNSUPPORTED
  assert lib.tapir_xconv_plan(ctx, 8, 80, 256, 1024, ctypes.byref(r), ctypes.byref(t), ctypes.byref(c)) == 0   # (wide form: one row of 80)
  assert (r.value, t.value, c.value) == (1, 8, 128)
  assert lib.tapir_xconv_plan(ctx, 8, 32, 128, 1024, ctypes.byref(r), ctypes.byref(t), ctypes.byref(c)) == _ffi.TAPIR_ERR_UNSUPPORTED
  assert lib.tapir_xconv_plan(ctx, 32, 32, 256, 1024, ctypes.byref(r), ctypes.byref(t), ctypes.byref(c)) == 0
  assert (r.value, t.value, c.value) == (4, 8, 128)      # the wide form: 4 rows x 32 per workgroup, chunks of 128 channels
  lib.tapir_destroy(ctx)


def test_xconv_rejects_a_pack_built_for_another_chunk_width():
  """The weight stream is ordered [chunk][tap][k-step][row tile] for ONE chunk width: using a pack made for a
  32-wide map (chunks of 256 channels) on a 64-wide map (chunks of 128) must be an error, not a wrong result."""
  lib = emu_lib()
  ctx = _ctx(lib)
  w = np.zeros((256, 256, 3, 3), np.float32)
  ws = ctypes.c_void_p()
  assert lib.tapir_xconv_pack(ctx, _p(w), 256, 256, 256, ctypes.byref(ws)) == 0
  x = np.zeros((1, 2, 64, 256), np.uint16); y = np.zeros((1, 2, 64, 256), np.uint16); b = np.zeros(256, np.float32)
  rc = lib.tapir_xconv(ctx, _p(x), ws, _p(b), None, _p(y), 1, 2, 64, 256, 256, 0, None)
  assert rc == _ffi.TAPIR_ERR_INVALID and b'chunks of 256' in lib.tapir_last_error(ctx)
  assert lib.tapir_xconv(ctx, _p(x), _p(w), _p(b), None, _p(y), 1, 2, 64, 256, 256, 0, None) == _ffi.TAPIR_ERR_INVALID
  lib.tapir_destroy(ctx)


def test_xconv_form_follows_the_clip_length():
  """tapir_xconv_plan_frames: the 128-pixel form only where the clip gives it enough workgroups (frames x small tiles x
  output-channel passes >= 512: the 1024 -> 256 convolution on a 32-wide map from 32 frames on, on a 64-wide one from 8), the 64-pixel form below; frames = 0 (unknown) is taken as
  many; maps only the wide form covers stay wide; and tapir_xconv_nt runs the form it is told to, refusing a pack built
  for the other one's chunk width."""
  lib = emu_lib()
  ctx = _ctx(lib)
  r, t, c, f = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  plan = lambda frames, H, W: (lib.tapir_xconv_plan_frames(ctx, frames, H, W, 1024, 256, ctypes.byref(r), ctypes.byref(t),
                                                           ctypes.byref(c), ctypes.byref(f)), r.value, t.value, c.value, f.value)
  assert plan(0, 32, 32) == (0, 4, 8, 128, 8) and plan(48, 32, 32) == (0, 4, 8, 128, 8) and plan(32, 32, 32)[4] == 8
  assert plan(31, 32, 32) == (0, 2, 16, 256, 4) and plan(4, 32, 32)[4] == 4
  assert plan(8, 64, 64) == (0, 2, 32, 128, 8) and plan(7, 64, 64) == (0, 1, 64, 128, 4)
  assert plan(1, 9, 72) == (0, 1, 9, 128, 8)                       # 72 cells per row: only the wide form
  # the 256 -> 1024 convolution runs four output-channel passes per tile: four times the workgroups, wide from 8 frames
  plan1 = lambda frames: (lib.tapir_xconv_plan_frames(ctx, frames, 32, 32, 256, 1024, ctypes.byref(r), ctypes.byref(t),
                                                      ctypes.byref(c), ctypes.byref(f)), f.value)
  assert plan1(8) == (0, 8) and plan1(7) == (0, 4)
  assert plan(-1, 32, 32)[0] == _ffi.TAPIR_ERR_INVALID
  # the same convolution through both forms, named explicitly
  rng = np.random.default_rng(5)
  N, H, W, C = 1, 4, 32, 256
  x = to_bf16_bits(_r(rng.standard_normal((N, H, W, C))))
  w = (rng.standard_normal((C, C, 3, 3)) / np.sqrt(9 * C)).astype(np.float32)
  b = (rng.standard_normal(C) * 0.1).astype(np.float32)
  ys = {}
  for form, cch in ((4, 256), (8, 128)):
    ws = ctypes.c_void_p()
    assert lib.tapir_xconv_pack(ctx, _p(w), C, C, cch, ctypes.byref(ws)) == 0
    y = np.zeros((N, H, W, C), np.uint16)
    assert lib.tapir_xconv_nt(ctx, _p(x), ws, _p(b), None, _p(y), N, H, W, C, C, 0, form, None) == 0, lib.tapir_last_error(ctx)
    other = 12 - form
    assert lib.tapir_xconv_nt(ctx, _p(x), ws, _p(b), None, _p(y), N, H, W, C, C, 0, other, None) == _ffi.TAPIR_ERR_INVALID
    ys[form] = from_bf16_bits(y)
    assert lib.tapir_conv_free(ctx, ws) == 0
  assert lib.tapir_xconv_nt(ctx, _p(x), None, _p(b), None, _p(x), N, H, W, C, C, 0, 5, None) == _ffi.TAPIR_ERR_INVALID
  np.testing.assert_allclose(ys[4], ys[8], atol=1.6e-2, rtol=1e-2)   # (different summation orders, bf16 outputs)
  assert np.abs(ys[4] - ys[8]).mean() < 1e-3 and np.abs(ys[8]).max() > 0.5
  lib.tapir_destroy(ctx)
