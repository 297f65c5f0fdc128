"""CPU: the oracle's operand-rounding mode (oracle.tapir_oracle.bf16_round, `rnd=`) against the bf16
instantiations of the HIP kernels compiled for the host emulator (tests/hipemu).  This pins WHERE the
bf16 build rounds -- the claim the `-m gpu` stage tests (tests/test_gpu_bf16_stages.py) rely on -- on
the build container: the emulated kernels execute the same arithmetic as the gfx950 ones, so they must
agree with the rounding oracle at accumulation-order noise, and several times closer than with the f32
oracle."""
import numpy as np
import pytest

from oracle import tapir_oracle as O
from tapnet_amd import _ffi, synthetic
from tests.emu_engine import EmuEngine


def _dev(a, b):
  d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).ravel()
  return float(d.max()), float(np.median(d))


def test_bf16_round_is_round_to_nearest_even():
  import torch
  x = np.random.default_rng(0).standard_normal(4096).astype(np.float32) * 37.0
  x[:4] = [1.00390625, 1.01171875, -1.00390625, 0.0]     # exact ties: to even
  ref = torch.tensor(x).bfloat16().float().numpy()
  np.testing.assert_array_equal(O.bf16_round(x), ref)
  assert O.bf16_round(x.reshape(64, 64)).shape == (64, 64)


@pytest.mark.parametrize('mode,T,N', [(2, 48, 2), (2, 20, 1), (3, 48, 3), (3, 70, 1), (1, 24, 2)])
def test_mixer_bf16_matches_rounding_oracle(mode, T, N):
  """fused (2), wide (3) and separate-launch (1) bf16 mixers: identical rounding points."""
  w = synthetic.make_weights(9, 1, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, pyramid_level=1, num_mixer_blocks=2, initial_resolution=(64, 64), dtype=_ffi.TAPIR_BF16)
  x = np.random.default_rng(T + N).standard_normal((N, T, 535)).astype(np.float32)
  assert e.lib.tapir_debug_set_mixer_mode(e.ctx, mode) == 0
  got = e.pips_mixer(x)
  ref16, _ = O.pips_mlp_mixer(w, x, num_blocks=2, rnd=O.bf16_round)
  ref32, _ = O.pips_mlp_mixer(w, x, num_blocks=2)
  mx16, md16 = _dev(got, ref16)
  mx32, md32 = _dev(got, ref32)
  print(mode, T, N, 'vs rounding oracle', mx16, md16, 'vs f32 oracle', mx32, md32)
  # measured: max 0.8-1.4e-3 (an operand on the other side of a bf16 rounding boundary), median <= 1.3e-5;
  # against the f32 oracle: max 4-5e-3, median 7e-4
  assert mx16 < 4e-3 and md16 < 1e-4, (mx16, md16)
  assert md16 < 0.1 * md32, (md16, md32)
  e.close()


def test_cost_volume_bf16_matches_rounding_oracle():
  w = synthetic.make_weights(3, 0, False, num_mixer_blocks=1, backbone=False)
  e = EmuEngine(w, pyramid_level=0, num_mixer_blocks=1, dtype=_ffi.TAPIR_BF16)
  rng = np.random.default_rng(5)
  Q, T = 20, 3
  grid = O.l2_normalize(rng.standard_normal((1, T, 32, 32, 256)).astype(np.float32))
  qp = synthetic.make_queries(6, Q, T, 256, 256)
  qf, _ = O.get_query_features([grid], [grid[..., :128]], [(256, 256)], qp, (1, T, 256, 256, 3))
  pts, occ, expd = e.tracks_from_cost_volume(qf[0], grid, qp)
  rp, ro, re, st = O.tracks_from_cost_volume(w, qf[0], grid, qp, (256, 256), 20.0, return_stages=True,
                                             rnd=O.bf16_round)
  _, ro32, re32 = O.tracks_from_cost_volume(w, qf[0], grid, qp, (256, 256), 20.0)
  ok = st['top2_rel_gap'] > 1e-3
  assert ok.mean() > 0.9
  print('points', _dev(pts[ok], rp[ok]), 'occ', _dev(occ, ro), 'vs f32', _dev(occ, ro32))
  assert _dev(pts[ok], rp[ok])[0] < 1e-3
  assert _dev(occ, ro)[0] < 1e-3 and _dev(expd, re)[0] < 1e-3      # (measured 9e-8)
  assert _dev(occ, ro)[1] < 0.5 * _dev(occ, ro32)[1]
  e.close()


@pytest.mark.parametrize('pyr', [0, 1])
def test_refine_pips_bf16_matches_rounding_oracle(pyr):
  w = synthetic.make_weights(8 + pyr, pyr, False, num_mixer_blocks=2, backbone=False)
  e = EmuEngine(w, pyramid_level=pyr, num_mixer_blocks=2, initial_resolution=(64, 64), dtype=_ffi.TAPIR_BF16)
  rng = np.random.default_rng(11 + pyr)
  Q, T, S = 5, 6, 64
  low = O.l2_normalize(rng.standard_normal((1, T, 8, 8, 256)).astype(np.float32))
  hi = O.l2_normalize(rng.standard_normal((1, T, 16, 16, 128)).astype(np.float32))
  qp = synthetic.make_queries(12, Q, T, S, S)
  ql, qh = O.get_query_features([low], [hi], [(S, S)], qp, (1, T, S, S, 3))
  queries, pyramid = [qh[0], ql[0]], [hi, low]
  for _ in range(pyr):
    queries.append(queries[-1]); pyramid.append(O.avg_pool_2x2(pyramid[-1]))
  pos = rng.uniform(-4, S + 4, (1, Q, T, 2)).astype(np.float32)
  occ = rng.standard_normal((1, Q, T)).astype(np.float32)
  expd = rng.standard_normal((1, Q, T)).astype(np.float32)
  last = None
  for it in range(2):
    out = e.refine_pips(queries, pyramid, pos, occ, expd, last, (S, S), (S, S))
    ref = O.refine_pips(w, queries, pyramid, pos, occ, expd, (S, S), last_iter=last, resize_hw=(S, S),
                        num_blocks=2, rnd=O.bf16_round)
    ref32 = O.refine_pips(w, queries, pyramid, pos, occ, expd, (S, S), last_iter=last, resize_hw=(S, S),
                          num_blocks=2)
    for k, name in enumerate(('pos', 'occ', 'expd', 'feats')):
      mx, md = _dev(out[k], ref[k])
      print(pyr, it, name, mx, md, 'vs f32', _dev(out[k], ref32[k]))
      assert mx < 5e-3 and md < 1e-4, (name, mx, md)      # (measured: max <= 1.7e-3, median <= 2e-6)
    pos, occ, expd, last = out
  e.close()
