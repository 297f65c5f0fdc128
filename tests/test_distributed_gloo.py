"""world_size-2 gloo test of the multi-GPU host layer (tapnet_amd/distributed.py):
frame-sharded features -> all-gather -> query-sharded tracking -> gather must equal
the single-process result.  The compute is a deterministic stand-in (the HIP path
needs a GPU); what is under test is the sharding / collective logic."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tapnet_amd import distributed as tdist
from tapnet_amd import tapir_model


class StubModel:
  """per-frame 'backbone' and per-query 'tracker' with the TAPIR method signatures and grid
  shapes ([B,T,H/8,W/8,256] and [B,T,H/4,W/4,128] at initial_resolution 16x16)."""
  initial_resolution = (16, 16)

  def get_feature_grids(self, video, is_training=False, refinement_resolutions=None):
    assert video.shape[1] > 0, 'the backbone must not run on an empty frame shard'
    m = video.mean(dim=(2, 3))   # [B,T,3]
    low = (m.sum(-1) * 3.0)[:, :, None, None, None].repeat(1, 1, 2, 2, 256)
    hi = video.amax(dim=(2, 3, 4))[:, :, None, None, None].repeat(1, 1, 4, 4, 128)
    return tapir_model.FeatureGrids((low, low), (hi, hi), ((16, 16), (16, 16)))

  def __call__(self, video, is_training, query_points, feature_grids=None, **kw):
    if feature_grids is None:
      feature_grids = self.get_feature_grids(video)
    assert query_points.shape[1] > 0, 'the tracker must not run on an empty query shard'
    T = video.shape[1]
    low, hi = feature_grids.lowres[1], feature_grids.hires[1]
    per_frame = low.sum(dim=(2, 3, 4)) + hi.sum(dim=(2, 3, 4))           # [B,T]: needs ALL frames
    tr = query_points[:, :, None, 1:] + per_frame[:, None, :, None]       # [B,Q,T,2]
    occ = query_points[:, :, None, 0] * per_frame[:, None, :]
    return dict(tracks=tr, occlusion=occ, expected_dist=-occ)


def _worker(rank, world, port, T, Q, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  g = torch.Generator().manual_seed(0)
  video = torch.rand(1, T, 16, 16, 3, generator=g)
  qp = torch.rand(1, Q, 3, generator=g) * 10
  out = tdist.sharded_call(StubModel(), video, qp)
  ref = StubModel()(video, False, qp)
  ok = all(out[k].shape == ref[k].shape and torch.allclose(out[k], ref[k], rtol=1e-5, atol=1e-3) for k in ref)
  if (T, Q) == (8, 6):   # the bf16 wire format: same result up to the rounding of the grids
    out16 = tdist.sharded_call(StubModel(), video, qp, grid_dtype=torch.bfloat16)
    ok = ok and all(torch.allclose(out16[k], ref[k], rtol=2e-2, atol=1e-2) for k in ref)
  q.put((rank, ok, tuple(out['tracks'].shape)))
  dist.destroy_process_group()


# even shards, ragged shards, and EMPTY shards (T < world, Q < world: that rank skips the compute
# but takes part in the collectives)
@pytest.mark.parametrize('T,Q', [(8, 6), (7, 5), (1, 1), (3, 1)])
def test_sharded_call_world2(T, Q):
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, T, Q, q)) for r in range(2)]
  for p in procs: p.start()
  res = [q.get(timeout=120) for _ in range(2)]
  for p in procs: p.join(30)
  for rank, ok, shape in res:
    assert ok, f'rank {rank} mismatch'
    assert shape == (1, Q, T, 2)


class StagedStubModel(StubModel):
  """Stand-in for a bf16 engine whose backbone writes the hot path's operand copies: the attributes
  tapnet_amd.distributed.gather_feature_grids looks at (dtype, _backbone._stage_ok, _staged, the _borrow /
  _global_frames keywords) and a __call__ that checks what it is handed."""
  dtype = 'bfloat16'

  class _BB:
    def _stage_ok(self, low):
      return low.shape[-1] == 256

  def __init__(self):
    self._backbone = self._BB()
    self._staged = []
    self.seen = None

  def get_feature_grids(self, video, is_training=False, refinement_resolutions=None, _borrow=False, _global_frames=None):
    fg = StubModel.get_feature_grids(self, video)
    if _borrow:
      b, t = video.shape[:2]
      low, hi = fg.lowres[0].reshape(b * t, 2, 2, 256), fg.hires[0].reshape(b * t, 4, 4, 128)
      low16 = low.to(torch.bfloat16)
      # tile order of a frame of 4 cells (one tile of 16, 12 of them padding): [tile][32 chunks][16 cells][8]
      tiled = torch.zeros(b * t, 1, 32, 16, 8, dtype=torch.bfloat16)
      tiled[:, 0, :, :4, :] = low16.reshape(b * t, 4, 32, 8).permute(0, 2, 1, 3)
      self._staged.append((low, low16, tiled.reshape(b * t, -1)))
      self._staged.append((hi, hi.to(torch.bfloat16), None))
    return fg

  def __call__(self, video, is_training, query_points, feature_grids=None, **kw):
    self.seen = feature_grids
    return StubModel.__call__(self, video, is_training, query_points, feature_grids=feature_grids, **kw)


def _staged_worker(rank, world, port, T, Q, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  g = torch.Generator().manual_seed(0)
  video = torch.rand(1, T, 16, 16, 3, generator=g)
  qp = torch.rand(1, Q, 3, generator=g) * 10
  m = StagedStubModel()
  out, fg = tdist.sharded_call(m, video, qp, grid_dtype=torch.bfloat16, return_grids=True)
  ok = isinstance(fg, tapir_model.StagedFeatureGrids) and len(fg.staged) == 2
  if ok:
    (lo32, lo16, lot), (hi32, hi16, hit) = fg.staged
    # what the whole clip gives on one rank
    whole = StagedStubModel()
    whole.get_feature_grids(video, _borrow=True)
    (_, wlo16, wlot), (_, whi16, _) = whole._staged
    ok = (hit is None and lo32.data_ptr() == fg.lowres[0].data_ptr() and hi32.data_ptr() == fg.hires[0].data_ptr()
          and torch.equal(lo32, lo16.float()) and torch.equal(hi32, hi16.float())
          and tuple(lo16.shape) == (1, T, 2, 2, 256) and tuple(lot.shape) == (1, T, 32 * 16 * 8)
          and torch.equal(lo16.reshape(wlo16.shape), wlo16) and torch.equal(lot.reshape(wlot.shape), wlot)
          and torch.equal(hi16.reshape(whi16.shape), whi16)
          and (m.seen is None or m.seen is fg))          # the hot path got the staged grids (None: empty query shard)
  ref = StubModel()(video, False, qp)
  ok = ok and all(torch.allclose(out[k], ref[k], rtol=2e-2, atol=1e-2) for k in ref)
  q.put((rank, bool(ok), tuple(out['tracks'].shape)))
  dist.destroy_process_group()


@pytest.mark.parametrize('T,Q', [(8, 6), (7, 5), (1, 1)])   # even, ragged and EMPTY frame / query shards
def test_sharded_call_gathers_staged_copies_world2(T, Q):
  """bf16 engine + bf16 wire: gather_feature_grids exchanges the operand copies the backbone wrote (row-major bf16, tile
  order, hi-res bf16) and returns StagedFeatureGrids keyed by the f32 grids it hands out; every gathered tensor equals
  what one rank computes for the whole clip (the collective logic with padded / empty shards of 2-D and 5-D tensors)."""
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_staged_worker, args=(r, 2, port, T, Q, q)) for r in range(2)]
  for p in procs: p.start()
  res = [q.get(timeout=120) for _ in range(2)]
  for p in procs: p.join(30)
  for rank, ok, shape in res:
    assert ok, f'rank {rank}: staged gather mismatch'
    assert shape == (1, Q, T, 2)


def test_shard_range_covers():
  for n in (1, 5, 48, 256):
    for w in (1, 2, 3, 8):
      spans = [tdist.shard_range(n, w, r) for r in range(w)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [e - s for s, e in spans]
      assert max(sizes) - min(sizes) <= 1
