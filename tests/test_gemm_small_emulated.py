"""The few-row GEMMs of the online model on the host emulator -- csrc/gemm.hpp gemm_small_kernel (whole K per workgroup,
the four waves split K and meet in LDS, operands global -> registers in fragment layout) and mlp_small_kernel (the channel MLP of
a block in ONE launch: a workgroup per (32 rows, 256 hidden units), the hidden tile in LDS, partial outputs that the next
mix_kernel / the final LayerNorm add while staging their rows): the causal T = 1 mixer (M = points x 1 frame rows) against the
oracle, and the three forms against each other (tapir_debug_set_gemm_mode), both element types, ragged row counts; and a
non-causal clip of few rows through the same path."""
import numpy as np
import pytest

from oracle import tapir_oracle as O
from tapnet_amd import _ffi, synthetic
from tests.emu_engine import EmuEngine


@pytest.mark.parametrize('dtype,N,pyr', [(_ffi.TAPIR_F32, 37, 1), (_ffi.TAPIR_F32, 256, 0), (_ffi.TAPIR_BF16, 70, 1),
                                         (_ffi.TAPIR_BF16, 1, 0)])
def test_online_mixer_small_gemm(dtype, N, pyr):
  w = synthetic.make_weights(5, pyr, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, pyramid_level=pyr, num_mixer_blocks=2, use_causal_conv=True, initial_resolution=(64, 64), dtype=dtype)
  rng = np.random.default_rng(N)
  x = rng.standard_normal((N, 1, 388 + 49 * (2 + pyr))).astype(np.float32)
  c1 = rng.standard_normal((2, N, 2, 512)).astype(np.float32)
  c2 = rng.standard_normal((2, N, 2, 2048)).astype(np.float32)
  outs = {}
  for mode in (3, 2, 1, 0):      # 3 (the engine's default): the persistent launch of csrc/mixer_online.hpp on the GPU; the emulator's
    assert e.lib.tapir_debug_set_gemm_mode(e.ctx, mode) == 0      # workgroups run eight at a time, so the host takes mode 2's launches
    outs[mode] = e.pips_mixer(x, c1, c2, get_ctx=True)
  assert np.array_equal(outs[3][0], outs[2][0]) and np.array_equal(outs[3][1][1], outs[2][1][1])
  ctx = {}
  for i in range(2):
    ctx[f'block_{i}_causal_1'] = c1[i]
    ctx[f'block_{i}_causal_2'] = c2[i]
  bf = dtype == _ffi.TAPIR_BF16
  ref, new_ctx = O.pips_mlp_mixer(w, x, num_blocks=2, use_causal_conv=True, causal_context=ctx, get_causal_context=True,
                                  rnd=O.bf16_round if bf else None)
  tol = 4e-3 if bf else 2e-4
  np.testing.assert_allclose(outs[1][0], ref, atol=tol)                # one-launch kernel vs the oracle
  np.testing.assert_allclose(outs[1][0], outs[0][0], atol=tol)         # vs the split-K pair
  np.testing.assert_allclose(outs[1][1][1], new_ctx['block_1_causal_1'], atol=tol)
  np.testing.assert_allclose(outs[2][0], ref, atol=tol)                # the one-launch MLP (default) vs the oracle
  np.testing.assert_allclose(outs[2][0], outs[1][0], atol=tol)
  np.testing.assert_allclose(outs[2][1][1], new_ctx['block_1_causal_1'], atol=tol)
  if N > 8:                                                            # (the other kernels really ran: another summation order;
    assert np.abs(outs[2][0] - outs[1][0]).max() > 0                   # one bf16 row can round to the same bits)
  e.close()


@pytest.mark.parametrize('dtype,N,T', [(_ffi.TAPIR_F32, 5, 9), (_ffi.TAPIR_BF16, 40, 7), (_ffi.TAPIR_BF16, 3, 11)])
def test_few_row_clip_through_the_one_launch_mlp(dtype, N, T):
  """A non-causal clip of few rows and fewer than 12 frames (a small query shard: tapnet_amd.distributed): mix_kernel with
  time chunks and halo rows reads the previous block's MLP pieces; three blocks (both residual buffers are read and written).
  tapir_model.py:33-156."""
  w = synthetic.make_weights(6, 1, False, num_mixer_blocks=3, backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_mixer_blocks=3, initial_resolution=(64, 64), dtype=dtype)
  rng = np.random.default_rng(N * 10 + T)
  x = rng.standard_normal((N, T, 388 + 49 * 3)).astype(np.float32)
  outs = {}
  for mode in (2, 1):
    assert e.lib.tapir_debug_set_gemm_mode(e.ctx, mode) == 0
    outs[mode] = e.pips_mixer(x)
  bf = dtype == _ffi.TAPIR_BF16
  ref, _ = O.pips_mlp_mixer(w, x, num_blocks=3, rnd=O.bf16_round if bf else None)
  tol = 6e-3 if bf else 3e-4
  np.testing.assert_allclose(outs[2], ref, atol=tol)
  np.testing.assert_allclose(outs[2], outs[1], atol=tol)
  assert np.abs(outs[2] - outs[1]).max() > 0
  e.close()
