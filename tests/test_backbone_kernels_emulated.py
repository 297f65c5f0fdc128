"""The backbone glue kernels (csrc/backbone.hpp: InstanceNorm statistics with the fused
residual add, normalise + ReLU with the SAME-padding border and the 2x2 subsampled copy,
per-pixel L2 normalisation) on the fiber emulator, against plain numpy."""
import ctypes

import numpy as np
import pytest

from tapnet_amd import _ffi
from tests.emu_engine import emu_lib
from tests.test_gemm_tiles_emulated import from_bf16_bits, to_bf16_bits


def _ctx(lib, dtype):
  cfg = _ffi.TapirCfg(1, 4, 1, 0, 20.0, 64, 64, dtype)
  ctx = ctypes.c_void_p()
  assert lib.tapir_create(ctypes.byref(ctx), ctypes.byref(cfg), 0) == 0
  return ctx


def _p(x):
  return None if x is None else x.ctypes.data_as(ctypes.c_void_p)


def _enc(x, dtype):
  return to_bf16_bits(x) if dtype == _ffi.TAPIR_BF16 else np.array(x, np.float32, copy=True)


def _dec(x, dtype):
  return from_bf16_bits(x) if dtype == _ffi.TAPIR_BF16 else x


@pytest.mark.parametrize('dtype', [_ffi.TAPIR_F32, _ffi.TAPIR_BF16])
@pytest.mark.parametrize('C,H,W,slabs', [(64, 8, 6, 3), (128, 4, 4, 1), (256, 6, 10, 7)])
def test_inorm_add_relu(dtype, C, H, W, slabs):
  lib = emu_lib()
  ctx = _ctx(lib, dtype)
  rng = np.random.default_rng(C + H)
  N = 2
  a = (rng.standard_normal((N, H, W, C)) * 2 + 3).astype(np.float32)   # mean >> 0: shifted sums
  b = rng.standard_normal((N, H, W, C)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, C).astype(np.float32)
  beta = rng.standard_normal(C).astype(np.float32)
  ea, eb = _enc(a, dtype), _enc(b, dtype)
  part = np.zeros((N, slabs, C, 2), np.float32)
  rc = lib.tapir_inorm_stats(ctx, _p(ea), _p(eb), _p(ea), _p(part), N, H * W, C, slabs, None)
  assert rc == 0, lib.tapir_last_error(ctx)
  x = _dec(ea, dtype)                       # the sum as stored (rounded for bf16)
  ref_sum = _dec(_enc(_dec(_enc(a, dtype), dtype) + _dec(_enc(b, dtype), dtype), dtype), dtype)
  np.testing.assert_array_equal(x, ref_sum)
  pad = H % 2 == 0 and W % 2 == 0
  oh, ow = (H + 1, W + 1) if pad else (H, W)
  y = _enc(np.zeros((N, oh, ow, C), np.float32), dtype)
  ys = _enc(np.zeros((N, H // 2, W // 2, C), np.float32), dtype) if pad else None
  ss = np.zeros((N, C, 2), np.float32)
  rc = lib.tapir_inorm_relu(ctx, _p(ea), _p(part), _p(gamma), _p(beta), _p(ss), _p(y), _p(ys), N, H, W, C,
                            slabs, 0, oh, ow, None)
  assert rc == 0, lib.tapir_last_error(ctx)
  mean = x.mean((1, 2), keepdims=True, dtype=np.float64)
  var = x.astype(np.float64).var((1, 2), keepdims=True)
  ref = np.maximum((x - mean) / np.sqrt(var + 1e-5) * gamma + beta, 0)
  got = _dec(y, dtype)
  tol = 3e-2 if dtype == _ffi.TAPIR_BF16 else 2e-5
  np.testing.assert_allclose(got[:, :H, :W], ref, atol=tol)
  if pad:
    assert (got[:, H:] == 0).all() and (got[:, :, W:] == 0).all()
    np.testing.assert_array_equal(_dec(ys, dtype), got[:, 0:H:2, 0:W:2])
  lib.tapir_destroy(ctx)


@pytest.mark.parametrize('dtype', [_ffi.TAPIR_F32, _ffi.TAPIR_BF16])
@pytest.mark.parametrize('C', [128, 256])
def test_l2_normalize(dtype, C):
  lib = emu_lib()
  ctx = _ctx(lib, dtype)
  rng = np.random.default_rng(C)
  x = rng.standard_normal((37, C)).astype(np.float32)
  x[5] = 0.0                                  # the max(., 1e-12) clamp
  ex = _enc(x, dtype)
  out = np.zeros((37, C), np.float32)
  rc = lib.tapir_l2_normalize(ctx, _p(ex), _p(out), 37, C, None)
  assert rc == 0, lib.tapir_last_error(ctx)
  xr = _dec(ex, dtype).astype(np.float64)
  ref = xr / np.sqrt(np.maximum((xr * xr).sum(-1, keepdims=True), 1e-12))
  np.testing.assert_allclose(out, ref, atol=1e-6)
  lib.tapir_destroy(ctx)
