"""The backbone's L2-normalise kernel writes the hot path's operand-type copies of the feature grids next to the f32
grids (tapir_l2_normalize_staged, csrc/backbone.hpp): row-major bf16 and -- for the 256-channel low-res map -- the
cost-volume kernel's tile order (csrc/pips.hpp PoolArgs::tiled, csrc/costvol_rows.hpp).  Emulated on the CPU: the
copies equal the round-to-nearest-even cast of the f32 output, element by element, in both layouts; and the
cost-volume stage reads the registered copies (tapir_set_staged_grid) to the same result as its own cast.
Reference arithmetic: tapnet/models/tapir_model.py:709-720 (per-pixel L2 normalisation), :399-471."""
import ctypes

import numpy as np
import pytest

from oracle import tapir_oracle as O
from tapnet_amd import _ffi, synthetic
from tests.emu_engine import EmuEngine


def _p(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def _bf16_bits(x):
  u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
  u = u + 0x7fff + ((u >> 16) & 1)
  return (u >> 16).astype(np.uint16)


@pytest.mark.parametrize('frames,h,w', [(2, 4, 8), (3, 5, 7)])
def test_l2norm_writes_the_operand_copies(frames, h, w):
  wts = synthetic.make_weights(3, 1, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(wts, num_mixer_blocks=1, initial_resolution=(8 * h, 8 * w), dtype=_ffi.TAPIR_BF16)
  rng = np.random.default_rng(frames + h)
  C, cells = 256, h * w
  x = _bf16_bits(rng.standard_normal((frames, h, w, C)).astype(np.float32))     # the backbone's bf16 activations
  out = np.zeros((frames, h, w, C), np.float32)
  op = np.zeros((frames, h, w, C), np.uint16)
  ntile = (cells + 15) // 16
  tiled = np.zeros((frames, ntile * 16 * C), np.uint16)
  rc = e.lib.tapir_l2_normalize_staged(e.ctx, _p(x), _p(out), _p(op), _p(tiled), frames * cells, C, cells, None)
  assert rc == 0, e.lib.tapir_last_error(e.ctx)
  xf = (x.astype(np.uint32) << 16).view(np.float32)
  ref = xf / np.sqrt(np.maximum((xf.astype(np.float64) ** 2).sum(-1, keepdims=True), 1e-12))
  np.testing.assert_allclose(out, ref, rtol=2e-6, atol=1e-7)
  assert np.array_equal(op, _bf16_bits(out))                                    # row-major copy = RNE cast of the f32 output
  t = tiled.reshape(frames, ntile, 32, 16, 8)                                    # [frame][tile][chunk][cell][8]
  flat = _bf16_bits(out).reshape(frames, cells, 32, 8)
  for f in range(frames):
    for cell in range(cells):
      assert np.array_equal(t[f, cell // 16, :, cell % 16, :], flat[f, cell])
    for cell in range(cells, ntile * 16):                                        # past the frame's end: never written
      assert not t[f, cell // 16, :, cell % 16, :].any()
  e.close()


def test_cost_volume_reads_registered_copies():
  """tapir_set_staged_grid: the stage uses the caller's copies instead of casting, with identical results."""
  wts = synthetic.make_weights(7, 1, False, num_mixer_blocks=1, backbone=False)
  h = wd = 12
  e = EmuEngine(wts, num_mixer_blocks=1, initial_resolution=(8 * h, 8 * wd), dtype=_ffi.TAPIR_BF16)
  rng = np.random.default_rng(2)
  B, Q, T = 1, 5, 2
  grid = O.l2_normalize(rng.standard_normal((B, T, h, wd, 256)).astype(np.float32))
  qf = O.l2_normalize(rng.standard_normal((B, Q, 256)).astype(np.float32))
  qp = np.stack([rng.integers(0, T, (B, Q)), rng.uniform(0, 8 * h, (B, Q)), rng.uniform(0, 8 * wd, (B, Q))], -1).astype(np.float32)
  base = e.tracks_from_cost_volume(qf, grid, qp)
  # the copies as the L2-normalise kernel writes them, from the same f32 grid
  cells, ntile = h * wd, (h * wd + 15) // 16
  op = _bf16_bits(grid).reshape(B * T, cells, 256)
  tiled = np.zeros((B * T, ntile * 16, 256), np.uint16)
  t5 = tiled.reshape(B * T, ntile, 16, 32, 8).transpose(0, 1, 3, 2, 4)           # view as [frame][tile][chunk][cell][8]
  tl = np.zeros((B * T, ntile, 32, 16, 8), np.uint16)
  for cell in range(cells):
    tl[:, cell // 16, :, cell % 16, :] = op[:, cell].reshape(B * T, 32, 8)
  del t5
  grid_c = np.ascontiguousarray(grid)
  assert e.lib.tapir_set_staged_grid(e.ctx, _p(grid_c), _p(op), _p(tl)) == 0
  # poison what the stage would otherwise cast from: only the registered copies can give the right answer
  pts = np.zeros((B, Q, T, 2), np.float32); occ = np.zeros((B, Q, T), np.float32); expd = np.zeros((B, Q, T), np.float32)
  rc = e.lib.tapir_tracks_from_cost_volume(e.ctx, _p(qf), _p(grid_c), _p(qp), B, Q, T, h, wd, _p(pts), _p(occ), _p(expd), None)
  assert rc == 0, e.lib.tapir_last_error(e.ctx)
  assert e.lib.tapir_clear_staged_grids(e.ctx) == 0
  assert np.array_equal(pts, base[0]) and np.array_equal(occ, base[1]) and np.array_equal(expd, base[2])
  e.close()
