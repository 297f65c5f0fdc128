"""Multi-GPU host layer: one process per GPU, ``torch.distributed`` (backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests and as the fallback when
two ranks share one device, which RCCL refuses).

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

The gather is ONE ``all_gather_into_tensor`` per tensor with the sharded axis leading
(for B = 1 the grids [1,T,h,w,C] are viewed as [T,h,w,C]: no transposes), straight into
the final buffer when the shards are even; ragged shards are padded to the largest one
and the valid rows compacted once.  Ranks whose shard is empty (T < world or Q < world)
skip the compute and still take part in the collectives.
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


def _needs_host_staging(x: torch.Tensor, group=None) -> bool:
  return x.is_cuda and dist.get_backend(group) == 'gloo'


def all_gather_rows(x: torch.Tensor, total: int, group=None) -> torch.Tensor:
  """All-gathers shards of a tensor that was split with shard_range along dim 0 ->
  [total, ...] on every rank.  x may have zero rows."""
  world = dist.get_world_size(group)
  if world == 1:
    return x
  x = x.contiguous()
  dev = x.device
  if _needs_host_staging(x, group):   # gloo has no device collectives: stage through the host
    return all_gather_rows(x.cpu(), total, group).to(dev)
  tail = tuple(x.shape[1:])
  if total % world == 0:
    out = torch.empty((total,) + tail, dtype=x.dtype, device=dev)
    dist.all_gather_into_tensor(out, x, group=group)
    return out
  big = -(-total // world)
  if x.shape[0] != big:
    pad = torch.zeros((big,) + tail, dtype=x.dtype, device=dev)
    pad[: x.shape[0]] = x
    x = pad
  buf = torch.empty((world * big,) + tail, dtype=x.dtype, device=dev)
  dist.all_gather_into_tensor(buf, x, group=group)
  # compact the valid rows of every rank's padded slab (one copy)
  rows = []
  for r in range(world):
    s, e = shard_range(total, world, r)
    rows.append(torch.arange(r * big, r * big + (e - s), device=dev))
  return buf.index_select(0, torch.cat(rows))


def all_gather_cat(x: torch.Tensor, dim: int, total: int, group=None) -> torch.Tensor:
  """All-gathers shards that were split with shard_range along `dim` and concatenates them.
  No transposes when every dimension before `dim` has size 1 (the B = 1 feature grids)."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return x
  lead = 1
  for d in x.shape[:dim]:
    lead *= d
  if dim == 0 or lead == 1:
    rows = all_gather_rows(x.reshape(x.shape[dim:]), total, group)
    return rows.reshape(tuple(x.shape[:dim]) + tuple(rows.shape))
  rows = all_gather_rows(x.movedim(dim, 0), total, group)
  return rows.movedim(0, dim).contiguous()


class ShapeOnly:
  """Stands in for the video when only its shape is needed (feature_grids given)."""

  def __init__(self, shape):
    self.shape = tuple(shape)


def _level_resolutions(model, video_hw, refinement_resolutions=None):
  """The resolutions get_feature_grids would produce, without running the backbone."""
  from tapnet_amd import model_utils
  if refinement_resolutions is None:
    refinement_resolutions = model_utils.generate_default_resolutions(
        tuple(video_hw), tuple(model.initial_resolution))
  return [tuple(model.initial_resolution)] + [tuple(int(v) for v in r) for r in refinement_resolutions]


def gather_feature_grids(model, video_local, num_frames: int, group=None,
                         grid_dtype: Optional[torch.dtype] = None):
  """Steps 1-2: backbone on the local frame slice, all-gather along T per distinct level.

  grid_dtype: element type on the wire (default: as computed, float32).  torch.bfloat16 halves
  the exchange (SURVEY.md 8e sizes it in bf16: 75 MB at config 2); the hot path of the bf16 build
  rounds the grids to bf16 anyway, only the query-feature sampling then sees rounded values."""
  from tapnet_amd import tapir_model
  B, t_local = video_local.shape[:2]
  lows, his = [], []
  # (the convolution implementation is chosen from the WHOLE clip's frame count, so a small shard runs
  # the kernels the unsharded call runs: bit-identical sharding, tapnet_amd/backbone.py)
  # (an engine-internal keyword: any other object with the reference's get_feature_grids works unchanged)
  # (ParameterizedTAPIR forwards get_feature_grids of the TAPIR it wraps: look through the wrapper, or a small
  # shard of a wrapped model would pick the per-shard convolution path and lose bit-identical sharding)
  inner = getattr(model, '_model', model)
  ours = hasattr(inner, '_backbone')
  # bf16 on the wire AND a bf16 engine whose backbone writes the hot path's operand copies (row-major bf16 + the
  # cost-volume kernel's tile order): gather THOSE and hand them to the hot path -- no
  # bf16 -> f32 -> bf16 round trip, no pool_cast_kernel pass over the gathered grids on every rank
  stage = bool(ours and grid_dtype == torch.bfloat16 and getattr(inner, 'dtype', None) == 'bfloat16'
               and inner._backbone is not None
               and inner._backbone._stage_ok(torch.empty((0, 1, 1, tapir_model.LOWRES_DIM))))
  staged_local = []
  if t_local > 0:
    kw = {'_global_frames': num_frames} if ours else {}
    if stage:
      inner._staged = []
      kw['_borrow'] = True
    fg = (inner if kw else model).get_feature_grids(video_local, **kw)
    res = tuple(fg.resolutions)
    levels = list(zip(fg.lowres, fg.hires))
    if stage:
      staged_local, inner._staged = list(inner._staged), []
      n_runs = sum(1 for i, r in enumerate(res) if i == 0 or tuple(r) != tuple(res[i - 1]))
      if len(staged_local) != 2 * n_runs:
        # (whether the backbone stages depends on the engine's dtype, the channel count and TAPIR_STAGE_GRIDS only --
        # identical on every rank; a silent per-rank fallback here would desynchronise the collectives below)
        raise RuntimeError('gather_feature_grids: the backbone wrote %d staged copies for %d backbone passes'
                           % (len(staged_local), n_runs))
  else:   # empty frame shard: contribute zero-length tensors of the right trailing shape
    res = tuple(_level_resolutions(model, video_local.shape[2:4]))
    dev = video_local.device
    levels = []
    for i, r in enumerate(res):
      if i == 0 or tuple(r) != tuple(res[i - 1]):
        cur = (torch.zeros((B, 0, r[0] // 8, r[1] // 8, 256), device=dev),
               torch.zeros((B, 0, r[0] // 4, r[1] // 4, 128), device=dev))
        if stage:
          cells = -(-((r[0] // 8) * (r[1] // 8)) // 16) * 16
          staged_local.append((cur[0], cur[0].to(torch.bfloat16),
                               torch.zeros((0, cells * 256), dtype=torch.bfloat16, device=dev)))
          staged_local.append((cur[1], cur[1].to(torch.bfloat16), None))
      levels.append(cur)
  # one exchange per DISTINCT level: consecutive levels of one resolution share their arrays
  # (get_feature_grids, tapir_model.py:666,722); keyed by position so that every rank -- empty
  # shards included -- issues the same sequence of collectives
  prev_res, pair = None, None
  staged_full, k = [], 0
  for (lo, hi), r in zip(levels, res):
    if pair is None or tuple(r) != prev_res:
      out = []
      if stage:
        (_, lo16, lo_t), (_, hi16, _) = staged_local[k], staged_local[k + 1]
        k += 2
        lo16f = all_gather_cat(lo16.reshape(B, t_local, *lo.shape[2:]), 1, num_frames, group)
        lotf = all_gather_cat(lo_t.reshape(B, t_local, lo_t.shape[-1]), 1, num_frames, group)   # (explicit: t_local may be 0)
        hi16f = all_gather_cat(hi16.reshape(B, t_local, *hi.shape[2:]), 1, num_frames, group)
        # the f32 grids the API hands out (query-feature sampling reads them; they key the staged copies)
        out = [lo16f.to(torch.float32), hi16f.to(torch.float32)]
        staged_full.append((out[0], lo16f, lotf))
        staged_full.append((out[1], hi16f, None))
      else:
        for g in (lo, hi):
          wire = g if grid_dtype is None else g.to(grid_dtype)
          full = all_gather_cat(wire, 1, num_frames, group)
          out.append(full if grid_dtype is None else full.to(torch.float32))
      pair, prev_res = tuple(out), tuple(r)
    lows.append(pair[0]); his.append(pair[1])
  if stage:
    fg = tapir_model.StagedFeatureGrids(tuple(lows), tuple(his), res)
    fg.staged = staged_full
    return fg
  return tapir_model.FeatureGrids(tuple(lows), tuple(his), res)


def sharded_call(model, video, query_points, group=None,
                 grid_dtype: Optional[torch.dtype] = None, return_grids: bool = False):
  """One clip over all ranks of `group`; every rank returns the full result.

  video [B,T,H,W,3] and query_points [B,Q,3] must be identical on every rank (each rank
  only reads its own frame / query slice).  return_grids: also return the gathered
  FeatureGrids (tests: the unsharded call on the same grids must give the same bits)."""
  world = dist.get_world_size(group) if dist.is_initialized() else 1
  rank = dist.get_rank(group) if dist.is_initialized() else 0
  if world == 1:
    if return_grids:
      fg = model.get_feature_grids(video)
      return model(video, False, query_points, feature_grids=fg), fg
    return model(video, False, query_points)
  B, T = video.shape[:2]
  Q = query_points.shape[1]
  t0, t1 = shard_range(T, world, rank)
  fg = gather_feature_grids(model, video[:, t0:t1], T, group, grid_dtype)
  q0, q1 = shard_range(Q, world, rank)
  keys = ('tracks', 'occlusion', 'expected_dist')
  if q1 > q0:
    out = model(ShapeOnly(video.shape), False, query_points[:, q0:q1], feature_grids=fg)
    local = {k: torch.as_tensor(out[k]) for k in keys}
  else:   # empty query shard: nothing to track, still part of the gather below
    dev = fg.lowres[0].device
    local = dict(tracks=torch.zeros((B, 0, T, 2), device=dev),
                 occlusion=torch.zeros((B, 0, T), device=dev),
                 expected_dist=torch.zeros((B, 0, T), device=dev))
  res = {k: all_gather_cat(local[k], 1, Q, group) for k in keys}
  return (res, fg) if return_grids else res
