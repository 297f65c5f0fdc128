"""Every tile shape / epilogue / operand type of the MFMA GEMM on the fiber emulator
(UNMODIFIED HIP source, tests/hipemu), through the tapir_debug_gemm hook of the C ABI:
ragged M and N edges, odd and even k-step counts, and the persistent multi-tile walk
(grid capped so that one workgroup computes several tiles)."""
import ctypes

import numpy as np
import pytest

from tapnet_amd import _ffi
from tests.emu_engine import emu_lib
from tests.test_kernels_emulated import bf16_round


def to_bf16_bits(x):
  u = np.ascontiguousarray(x, np.float32).view(np.uint32)
  return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def from_bf16_bits(b):
  return (b.astype(np.uint32) << 16).view(np.float32)


def gelu_tanh(x):
  return 0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3)))


def run_gemm(lib, dtype, A, W, bias, resid, epi, tile, max_grid=0):
  cfg = _ffi.TapirCfg(1, 4, 1, 0, 20.0, 64, 64, dtype)
  ctx = ctypes.c_void_p()
  assert lib.tapir_create(ctypes.byref(ctx), ctypes.byref(cfg), 0) == 0
  M, K = A.shape
  N = W.shape[0]
  if dtype == _ffi.TAPIR_BF16:
    a, w = to_bf16_bits(A), to_bf16_bits(W)
  else:
    a, w = np.ascontiguousarray(A, np.float32), np.ascontiguousarray(W, np.float32)
  out_bf = dtype == _ffi.TAPIR_BF16 and epi == 1
  C = np.zeros((M, N), np.uint16 if out_bf else np.float32)
  p = lambda x: None if x is None else x.ctypes.data_as(ctypes.c_void_p)
  rc = lib.tapir_debug_gemm(ctx, p(a), K, p(w), K, p(bias), p(resid), N, p(C), N, M, N, K, epi,
                            tile | (max_grid << 8), None)
  assert rc == 0, lib.tapir_last_error(ctx)
  lib.tapir_destroy(ctx)
  return from_bf16_bits(C) if out_bf else C


@pytest.mark.parametrize('dtype', [_ffi.TAPIR_F32, _ffi.TAPIR_BF16])
@pytest.mark.parametrize('tile', [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
@pytest.mark.parametrize('epi', [0, 1, 2])
def test_gemm_tile_shapes(dtype, tile, epi):
  lib = emu_lib()
  rng = np.random.default_rng(10 * tile + epi)
  kstep = 64 if dtype == _ffi.TAPIR_BF16 else 32
  M, N, K = 200, 132, 3 * kstep          # ragged in M and N for every tile, odd k-step count
  A = rng.standard_normal((M, K)).astype(np.float32)
  W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
  bias = rng.standard_normal(N).astype(np.float32)
  resid = rng.standard_normal((M, N)).astype(np.float32)
  out = run_gemm(lib, dtype, A, W, bias, resid, epi, tile)
  if dtype == _ffi.TAPIR_BF16:
    A, W = bf16_round(A), bf16_round(W)
  ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
  if epi == 1:
    ref = gelu_tanh(ref)
  if epi == 2:
    ref = ref + resid
  tol = 2e-2 if (dtype == _ffi.TAPIR_BF16 and epi == 1) else 2e-4
  np.testing.assert_allclose(out, ref, atol=tol)


@pytest.mark.parametrize('tile', [1, 2, 3, 4, 5, 6, 7, 8, 9, 10])
def test_gemm_persistent_walk(tile):
  """8 workgroups walk 12..28 tiles: the next tile's first DMA is issued before the epilogue."""
  lib = emu_lib()
  rng = np.random.default_rng(tile)
  M, N, K = 400, 260, 128
  A = rng.standard_normal((M, K)).astype(np.float32)
  W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
  out = run_gemm(lib, _ffi.TAPIR_F32, A, W, None, None, 0, tile, max_grid=8)
  ref = A.astype(np.float64) @ W.astype(np.float64).T
  np.testing.assert_allclose(out, ref, atol=2e-4)
