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


def test_shard_range_covers():
  for n in (1, 5, 48, 256):
    for w in (1, 2, 3, 8):
      spans = [tdist.shard_range(n, w, r) for r in range(w)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      sizes = [e - s for s, e in spans]
      assert max(sizes) - min(sizes) <= 1
