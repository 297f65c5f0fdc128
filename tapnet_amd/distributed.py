"""Multi-GPU host layer: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no multi-device inference path; the path
shards along two independent axes:

* clips (batch): embarrassingly parallel, no exchange -- ``bench.py --shard clips``
  simply gives every rank its own clip (weak scaling).
* one clip over N GPUs (``sharded_call``):
    1. frames: the backbone is per-frame independent (InstanceNorm per image,
       tapir_model.py:325) -> rank r computes feature grids for its frame slice;
    2. ONE all-gather per distinct feature level along T (the only exchange: every
       query needs all frames, because the mixer convolves along time :59-82);
    3. queries: trajectories are independent per query
       (tapnet/tapvid/README.md:32-38) -> rank r runs the HIP hot path for its
       query slice against the full grids;
    4. all-gather of the (tiny) per-query outputs.
  With 7 direct xGMI links per GPU each peer's shard travels its own link, so the
  all-gather is a single direct exchange rather than a ring.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, world: int, rank: int) -> Tuple[int, int]:
  """Contiguous, balanced split of range(n): the first n % world ranks get one extra."""
  base, rem = divmod(n, world)
  start = rank * base + min(rank, rem)
  return start, start + base + (1 if rank < rem else 0)


def all_gather_cat(x: torch.Tensor, dim: int, total: int, group=None) -> torch.Tensor:
  """All-gathers shards that were split with shard_range along `dim` and concatenates them."""
  world = dist.get_world_size(group)
  if world == 1:
    return x
  x = x.contiguous()
  if total % world == 0 and dim == 0:
    out = torch.empty((total,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out
  # general case: move `dim` first, pad to the largest shard, gather, trim
  xm = x.movedim(dim, 0).contiguous()
  big = -(-total // world)
  pad = torch.zeros((big,) + tuple(xm.shape[1:]), dtype=x.dtype, device=x.device)
  pad[: xm.shape[0]] = xm
  parts = [torch.empty_like(pad) for _ in range(world)]
  dist.all_gather(parts, pad, group=group)
  outs = []
  for r, p in enumerate(parts):
    s, e = shard_range(total, world, r)
    outs.append(p[: e - s])
  return torch.cat(outs, 0).movedim(0, dim).contiguous()


class ShapeOnly:
  """Stands in for the video when only its shape is needed (feature_grids given)."""

  def __init__(self, shape):
    self.shape = tuple(shape)


def gather_feature_grids(model, video_local, num_frames: int, group=None):
  """Steps 1-2: backbone on the local frame slice, all-gather along T per distinct level."""
  from tapnet_amd import tapir_model
  fg = model.get_feature_grids(video_local)
  lows, his, cache = [], [], {}
  for lo, hi in zip(fg.lowres, fg.hires):
    key = (lo.data_ptr() if hasattr(lo, 'data_ptr') else id(lo))
    if key not in cache:
      cache[key] = (all_gather_cat(lo, 1, num_frames, group), all_gather_cat(hi, 1, num_frames, group))
    lows.append(cache[key][0]); his.append(cache[key][1])
  return tapir_model.FeatureGrids(tuple(lows), tuple(his), fg.resolutions)


def sharded_call(model, video, query_points, group=None) -> Dict[str, Any]:
  """One clip over all ranks of `group`; every rank returns the full result.

  video [B,T,H,W,3] and query_points [B,Q,3] must be identical on every rank (each rank
  only reads its own frame / query slice)."""
  world = dist.get_world_size(group) if dist.is_initialized() else 1
  rank = dist.get_rank(group) if dist.is_initialized() else 0
  if world == 1:
    return model(video, False, query_points)
  B, T = video.shape[:2]
  Q = query_points.shape[1]
  t0, t1 = shard_range(T, world, rank)
  fg = gather_feature_grids(model, video[:, t0:t1], T, group)
  q0, q1 = shard_range(Q, world, rank)
  out = model(ShapeOnly(video.shape), False, query_points[:, q0:q1], feature_grids=fg)
  res = {}
  for k in ('tracks', 'occlusion', 'expected_dist'):
    res[k] = all_gather_cat(torch.as_tensor(out[k]), 1, Q, group)
  return res
