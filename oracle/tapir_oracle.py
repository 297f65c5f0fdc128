"""TEST INFRASTRUCTURE ONLY -- numpy CPU restatement of the TAPIR inference hot path.

This file is the *oracle* (checker) for the HIP kernels in ``tapnet_amd/csrc``.
It restates, in plain numpy, the algorithm of the reference
(google-deepmind/tapnet, JAX model ``tapnet/models/tapir_model.py`` with its
PyTorch twin ``tapnet/torch/*`` used to pin it).  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Nothing under ``tapnet_amd/`` imports it; the product path has no CPU fallback.

Parity status: the reference ships NO tests / golden vectors for this path
(SURVEY.md section 4).  The oracle is therefore pinned against OUTPUTS OF THE
REFERENCE ITSELF: ``oracle/make_golden.py`` imports the reference PyTorch TAPIR
from /root/reference, runs it on seeded inputs and commits stage-boundary
tensors under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this
file against those fixtures (and, when /root/reference is present, against the
live reference).  JAX / Haiku cannot be installed offline; JAX-vs-torch
differences listed in SURVEY.md section 8c are resolved in favour of the JAX
source text (e.g. per-axis normalisation in ``interp``), and since round 3 the
JAX text itself is EXECUTED over numpy stand-ins for jax / haiku
(``oracle/hk_numpy_shim.py``; ``oracle/make_jax_golden.py`` ->
``tests/golden/jax_*.npz``; ``tests/test_jax_reference_pin.py``): non-square
clips, the multi-resolution path with the antialiased resize, the online causal
loop and the Haiku parameter tree agree with this file at 4e-5 px / 3e-6.

Weights are a flat ``dict[str, np.ndarray]`` keyed by the reference's torch
``state_dict`` names (tapnet/torch/tapir_model.py:115-137).

All array layouts follow the reference: grids are ``[B, T, h, w, C]``
(channels last), query points are ``(t, y, x)``, tracks are ``(x, y)``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

HIRES_DIM = 128  # tapir_model.py:320
LOWRES_DIM = 256  # tapir_model.py:321


# ---------------------------------------------------------------------------
# operand rounding of the bf16 build (test infrastructure for the bf16 HIP kernels)
# ---------------------------------------------------------------------------
def bf16_round(x):
  """float32 -> nearest-even bfloat16 -> float32 (the rounding of the HIP kernels' pack_bf16x2 /
  host_f2bf).  The functions below take `rnd=None | bf16_round`: with rnd they round exactly the
  tensors the bf16 build of the engine rounds -- the operands of its MFMA products (GEMM weights,
  mixer input rows, LayerNorm / GELU outputs that feed a GEMM, feature grids and query vectors of the
  cost-volume einsum, the 16-channel map and kernel of the occlusion convolution) -- and keep everything
  else (residual stream, LayerNorm, temporal convolutions, softmax / soft arg max, biases) in f32, so a
  bf16 kernel can be held to this restatement at accumulation-order noise instead of to a loose
  drift bound.  rnd=None is the reference's f32 arithmetic, bit for bit what the fixtures pin."""
  u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
  u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
  return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def _id(x):
  return x


# ---------------------------------------------------------------------------
# plumbing (R9)
# ---------------------------------------------------------------------------
def convert_grid_coordinates(coords, input_grid_size, output_grid_size):
  """coords * out / in  (tapnet/utils/transforms.py:75-76)."""
  dt = coords.dtype
  return (coords * np.asarray(output_grid_size, dtype=dt)
          / np.asarray(input_grid_size, dtype=dt)).astype(dt)


def generate_default_resolutions(full_size, train_size, num_levels=None):
  """tapnet/utils/model_utils.py:317-359."""
  if all(x == y for x, y in zip(train_size, full_size)):
    return [tuple(train_size)]
  if num_levels is None:
    size_ratio = np.array(full_size) / np.array(train_size)
    num_levels = int(np.ceil(np.max(np.log2(size_ratio))) + 1)
  if num_levels <= 1:
    return [tuple(train_size)]
  h, w = full_size[0:2]
  ll_h, ll_w = train_size[0:2]
  sizes = []
  for i in range(num_levels):
    sizes.append((
        int(round((ll_h * (h / ll_h) ** (i / (num_levels - 1))) // 8)) * 8,
        int(round((ll_w * (w / ll_w) ** (i / (num_levels - 1))) // 8)) * 8,
    ))
  return sizes


def postprocess_occlusions(occlusions, expected_dist):
  """visible iff (1-sigmoid(occ))(1-sigmoid(expd)) > 0.5 (model_utils.py:376-389)."""
  sig = lambda v: 1.0 / (1.0 + np.exp(-v))
  return (1 - sig(occlusions)) * (1 - sig(expected_dist)) > 0.5


def l2_normalize(x):
  """x / sqrt(max(sum(x^2), 1e-12)) over channels (tapir_model.py:709-720)."""
  s = np.sum(np.square(x), axis=-1, keepdims=True)
  return x / np.sqrt(np.maximum(s, np.asarray(1e-12, x.dtype)))


def avg_pool_2x2(grid):
  """hk.avg_pool(grid, [1,1,2,2,1], [1,1,2,2,1], 'VALID') (tapir_model.py:995-1000)."""
  b, t, h, w, c = grid.shape
  h2, w2 = h // 2, w // 2
  g = grid[:, :, : h2 * 2, : w2 * 2]
  g = g.reshape(b, t, h2, 2, w2, 2, c)
  return g.mean(axis=(3, 5), dtype=grid.dtype)


# ---------------------------------------------------------------------------
# R8: query features -- trilinear sample, clamp at the borders
# ---------------------------------------------------------------------------
def interp_nearest_3d(vol, pts_tyx):
  """model_utils.interp(mode='nearest') on a [T,h,w,C] volume (model_utils.py:177-206).

  t is NOT shifted, y and x are shifted by -0.5; order-1 map_coordinates with
  index clamping (jax.scipy.ndimage.map_coordinates 'nearest').
  pts_tyx: [N,3] in grid units.  Returns [N,C].
  """
  dt = vol.dtype
  c = pts_tyx.astype(dt).copy()
  c[:, 1:] -= dt.type(0.5)
  sizes = vol.shape[:3]
  lo = np.floor(c)
  wu = (c - lo).astype(dt)
  wl = (dt.type(1) - wu).astype(dt)
  lo = lo.astype(np.int64)
  out = np.zeros((c.shape[0], vol.shape[3]), dtype=dt)
  for dt_ in (0, 1):
    it = np.clip(lo[:, 0] + dt_, 0, sizes[0] - 1)
    wt = wu[:, 0] if dt_ else wl[:, 0]
    for dy in (0, 1):
      iy = np.clip(lo[:, 1] + dy, 0, sizes[1] - 1)
      wy = wu[:, 1] if dy else wl[:, 1]
      for dx in (0, 1):
        ix = np.clip(lo[:, 2] + dx, 0, sizes[2] - 1)
        wx = wu[:, 2] if dx else wl[:, 2]
        out += vol[it, iy, ix, :] * (wt * wy * wx)[:, None]
  return out


def get_query_features(lowres: Sequence[np.ndarray], hires: Sequence[np.ndarray],
                       resolutions: Sequence[Tuple[int, int]],
                       query_points: np.ndarray, video_shape) -> Tuple[list, list]:
  """TAPIR.get_query_features (tapir_model.py:731-856).

  video_shape = [B,T,H,W,3].  Returns (lowres_q [B,Q,256] per level,
  hires_q [B,Q,128] per level).
  """
  shape = video_shape
  q_low, q_hi = [], []
  curr = (-1, -1)
  for i, res in enumerate(resolutions):
    if tuple(curr) == tuple(res):
      q_low.append(q_low[-1])
      q_hi.append(q_hi[-1])
      continue
    curr = res
    outs = []
    for grid in (lowres[i], hires[i]):
      pos = convert_grid_coordinates(
          query_points.astype(grid.dtype), shape[1:4], grid.shape[1:4])
      outs.append(np.stack([
          interp_nearest_3d(grid[b], pos[b]) for b in range(grid.shape[0])]))
    q_low.append(outs[0])
    q_hi.append(outs[1])
  return q_low, q_hi


# ---------------------------------------------------------------------------
# R2 / R5: cost volume -> tracks
# ---------------------------------------------------------------------------
def build_cost_volume(interp_feature, feature_grid):
  """einsum('bnc,bthwc->tbnhw') (tapir_model.py:433)."""
  return np.einsum('bnc,bthwc->tbnhw', interp_feature, feature_grid,
                   optimize=True).astype(feature_grid.dtype)


def _same_pads(n, k, s):
  """XLA 'SAME' padding: extra pad goes on the high side."""
  out = -(-n // s)
  total = max((out - 1) * s + k - n, 0)
  lo = total // 2
  return lo, total - lo, out


def conv2d_same(x, w, b, stride=1):
  """NHWC conv, XLA SAME padding, cross-correlation.

  x [N,H,W,Ci]; w [Co,Ci,kh,kw] (torch layout); b [Co].
  """
  co, ci, kh, kw = w.shape
  n, h, ww, _ = x.shape
  ply, phy, oh = _same_pads(h, kh, stride)
  plx, phx, ow = _same_pads(ww, kw, stride)
  xp = np.pad(x, ((0, 0), (ply, phy), (plx, phx), (0, 0)))
  out = np.zeros((n, oh, ow, co), dtype=x.dtype)
  for i in range(kh):
    for j in range(kw):
      patch = xp[:, i: i + (oh - 1) * stride + 1: stride,
                 j: j + (ow - 1) * stride + 1: stride, :]
      out += patch @ w[:, :, i, j].T.astype(x.dtype)
  return out + b.astype(x.dtype)


def soft_argmax_heatmap(softmax_val, threshold=5.0):
  """model_utils.soft_argmax_heatmap (model_utils.py:209-247), batched over [...,h,w].

  Returns (points [...,2] as (x,y) in grid units, argmax flat index [...]).
  """
  dt = softmax_val.dtype
  h, w = softmax_val.shape[-2:]
  xs, ys = np.meshgrid(np.arange(w), np.arange(h))
  coords = np.stack([xs + 0.5, ys + 0.5], axis=-1).astype(dt)  # [h,w,2]
  flat = softmax_val.reshape(softmax_val.shape[:-2] + (h * w,))
  amax = np.argmax(flat, axis=-1)  # first maximum, as jnp.argmax
  pos = coords.reshape(-1, 2)[amax]  # [...,2]
  d2 = np.sum(np.square(coords - pos[..., None, None, :]), axis=-1)
  valid = (d2 < dt.type(threshold) ** 2).astype(dt)
  wsum = np.sum(coords * (valid * softmax_val)[..., None], axis=(-3, -2))
  sw = np.maximum(np.sum(valid * softmax_val, axis=(-2, -1)), dt.type(1e-12))
  return (wsum / sw[..., None]).astype(dt), amax


def heatmaps_to_points(all_pairs_softmax, image_hw, query_points=None,
                       threshold=5.0):
  """model_utils.heatmaps_to_points (model_utils.py:250-314).

  all_pairs_softmax [B,N,T,h,w]; image_hw = (H, W) of im_shp; query_points
  [B,N,3] (t,y,x) in im_shp coordinates.  Returns [B,N,T,2] (x,y).
  """
  dt = all_pairs_softmax.dtype
  pts, _ = soft_argmax_heatmap(all_pairs_softmax, threshold)
  b, n, t, h, w = all_pairs_softmax.shape
  pts = convert_grid_coordinates(pts, (w, h), (image_hw[1], image_hw[0]))
  if query_points is not None:
    qf = np.round(query_points[..., 0].astype(dt)).astype(np.int32)  # half-even
    is_q = (qf[:, :, None] == np.arange(t, dtype=np.int32)[None, None, :])
    is_q = is_q[..., None]
    qxy = query_points[:, :, None, 2:0:-1].astype(dt)
    pts = pts * (1 - is_q).astype(dt) + qxy * is_q.astype(dt)
  return pts.astype(dt)


def tracks_from_cost_volume(weights: Dict[str, np.ndarray], interp_feature,
                            feature_grid, query_points, im_hw=(256, 256),
                            softmax_temperature=20.0, return_stages=False, rnd=None):
  """TAPIR.tracks_from_cost_volume (tapir_model.py:399-471).

  interp_feature [B,N,C]; feature_grid [B,T,h,w,C]; query_points [B,N,3] in
  im_hw coordinates or None.  Returns points [B,N,T,2], occ [B,N,T],
  expd [B,N,T] (+ dict of stage tensors).
  """
  p = 'torch_cost_volume_track_mods.'
  rnd = rnd or _id
  dt = feature_grid.dtype
  cv = build_cost_volume(rnd(interp_feature), rnd(feature_grid))  # [T,B,N,h,w]
  t, b, n, h, w = cv.shape
  x = cv.reshape(t * b * n, h, w, 1)
  hid1 = np.maximum(conv2d_same(x, weights[p + 'hid1.weight'],
                                weights[p + 'hid1.bias']), 0)
  logits = conv2d_same(hid1, weights[p + 'hid2.weight'], weights[p + 'hid2.bias'])
  logits = logits.reshape(t, b, n, h, w).transpose(1, 2, 0, 3, 4)  # b n t h w
  z = logits * dt.type(softmax_temperature)
  z = z - z.max(axis=(-2, -1), keepdims=True)
  e = np.exp(z)
  sm = (e / e.sum(axis=(-2, -1), keepdims=True)).astype(dt)
  points = heatmaps_to_points(sm, im_hw, query_points)

  occ = np.maximum(conv2d_same(rnd(hid1), rnd(weights[p + 'hid3.weight']),
                               weights[p + 'hid3.bias'], stride=2), 0)
  occ = occ.mean(axis=(1, 2), dtype=dt)
  occ = np.maximum(occ @ weights[p + 'hid4.weight'].T.astype(dt)
                   + weights[p + 'hid4.bias'].astype(dt), 0)
  occ = occ @ weights[p + 'occ_out.weight'].T.astype(dt) \
      + weights[p + 'occ_out.bias'].astype(dt)
  occ = occ.reshape(t, b, n, 2).transpose(1, 2, 0, 3)
  occlusion, expected_dist = occ[..., 0], occ[..., 1]
  if return_stages:
    flat = sm.reshape(b, n, t, h * w)
    srt = np.sort(flat, axis=-1)
    stages = dict(cost_volume=cv, logits=logits, softmax=sm,
                  argmax=np.argmax(flat, axis=-1),
                  top2_rel_gap=(srt[..., -1] - srt[..., -2]) / srt[..., -1])
    return points, occlusion, expected_dist, stages
  return points, occlusion, expected_dist


def tapnet_tracks_from_cost_volume(weights: Dict[str, np.ndarray], interp_feature,
                                  feature_grid, query_points, im_hw=(256, 256),
                                  softmax_temperature=10.0, return_stages=False):
  """TAPNet.tracks_from_cost_volume (tapnet/models/tapnet_model.py:111-171).

  The TAP-Net head is the TAPIR head with three differences, all visible in the reference text:
  there is NO ReLU between the stride-2 convolution `hid3` and the spatial mean (:160-161 vs
  tapir_model.py:460-462), `occ_out` has ONE output (occlusion only, no expected distance; :90),
  and the softmax temperature is 10 (:61).  Conv3D kernels of shape [1,3,3] over (t, h, w) are 2-D
  3x3 convolutions per frame.  Weights: `tapnet_cost_volume_track_mods.{hid1,hid2,hid3,hid4,
  occ_out}.{weight,bias}` in the torch layout of the TAPIR head (hid1 [16,heads,3,3], ..., occ_out [1,16]).
  num_heads = hid1's input channels: channel c of the features belongs to head c % heads
  ('b t h w (c d) -> b t h w c d', :247-254) and every head's cost map is one input channel of hid1.

  Parity status: pinned to the reference's OWN code executed over numpy stand-ins for jax / haiku
  (oracle/hk_numpy_shim.py, oracle/make_tapnet_golden.py -> tests/golden/tapnet_head.npz,
  tests/test_tapnet_reference_pin.py: tracks 2e-5 px, occlusion 1e-8) -- the reference has no torch twin
  of TAP-Net and JAX cannot run offline, so the primitives (convolution padding, softmax,
  map_coordinates) under the reference's lines are restated ones.

  interp_feature [B,N,C]; feature_grid [B,T,h,w,C]; returns points [B,N,T,2], occlusion [B,N,T]."""
  p = 'tapnet_cost_volume_track_mods.'
  dt = feature_grid.dtype
  heads = weights[p + 'hid1.weight'].shape[1]
  cv = np.stack([build_cost_volume(interp_feature[..., d::heads], feature_grid[..., d::heads])
                 for d in range(heads)], -1)              # [T,B,N,h,w,heads]  ('bncd,bthwcd->tbnhwd')
  t, b, n, h, w, _ = cv.shape
  x = cv.reshape(t * b * n, h, w, heads)
  hid1 = np.maximum(conv2d_same(x, weights[p + 'hid1.weight'], weights[p + 'hid1.bias']), 0)
  logits = conv2d_same(hid1, weights[p + 'hid2.weight'], weights[p + 'hid2.bias'])
  logits = logits.reshape(t, b, n, h, w).transpose(1, 2, 0, 3, 4)  # b n t h w
  z = logits * dt.type(softmax_temperature)
  z = z - z.max(axis=(-2, -1), keepdims=True)
  e = np.exp(z)
  sm = (e / e.sum(axis=(-2, -1), keepdims=True)).astype(dt)
  points = heatmaps_to_points(sm, im_hw, query_points)
  occ = conv2d_same(hid1, weights[p + 'hid3.weight'], weights[p + 'hid3.bias'], stride=2)   # no ReLU
  occ = occ.mean(axis=(1, 2), dtype=dt)
  occ = np.maximum(occ @ weights[p + 'hid4.weight'].T.astype(dt) + weights[p + 'hid4.bias'].astype(dt), 0)
  occ = occ @ weights[p + 'occ_out.weight'].T.astype(dt) + weights[p + 'occ_out.bias'].astype(dt)
  occlusion = occ.reshape(t, b, n).transpose(1, 2, 0)
  if return_stages:
    flat = sm.reshape(b, n, t, h * w)
    srt = np.sort(flat, axis=-1)
    return points, occlusion, dict(top2_rel_gap=(srt[..., -1] - srt[..., -2]) / srt[..., -1])
  return points, occlusion


def cycle_consistency_tracks(query_feats, feature_grid, query_points, im_hw, softmax_temperature=10.0,
                             dist_threshold=48.0, rnd=None, return_stages=False):
  """The forward-backward cycle-consistency tracker of the TAP-Net evaluation path
  (tapnet/training/supervised_point_prediction.py:443-546): no learned head at all --
    forward : tracks = heatmaps_to_points(softmax(temperature * einsum('bnc,bthwc->bnthw')), query_points)  :453-469
    features: bilinear samples of the grid at the tracked point of every frame (model_utils.interp,
              (t, y, x) -> grid coordinates, t an exact frame)                                                 :473-496
    backward: the sampled vector of (query n, frame t) against the grid of the QUERY's frame
              (round(t_query)), einsum('bntc,bnhwc->bnthw'), the same soft arg max without the override       :501-531
    occluded: the backward point lands more than 48 px from the query -> logit +10, else -10                  :533-539
  Line :537 of the reference indexes the query points as `query_points[jnp.newaxis, 2:0:-1]`, which slices the
  BATCH axis (and cannot broadcast against [b,n,t,2]); the evident intent -- the query's (x, y), i.e.
  query_points[:, :, None, 2:0:-1] -- is what is restated here.  Everything before that line is pinned to the
  reference's own lines executed over numpy stand-ins (oracle/make_cycle_golden.py).

  query_feats [B,N,C], feature_grid [B,T,h,w,C], query_points [B,N,3] (t,y,x) in im_hw pixels.
  Returns tracks [B,N,T,2] (x,y) px, occlusion logits [B,N,T] (and the inverse tracks)."""
  rnd = rnd or _id
  dt = feature_grid.dtype
  b, t, h, w, c = feature_grid.shape
  n = query_feats.shape[1]

  def soft_points(dots, qp):
    z = dots * dt.type(softmax_temperature)
    z = z - z.max(axis=(-2, -1), keepdims=True)
    e = np.exp(z)
    return heatmaps_to_points((e / e.sum(axis=(-2, -1), keepdims=True)).astype(dt), im_hw, qp)

  fg = rnd(feature_grid)
  dots = np.einsum('bnc,bthwc->bnthw', rnd(query_feats), fg, optimize=True).astype(dt)
  tracks = soft_points(dots, query_points)
  # position_in_grid = (frame, y, x) * grid_shape / im_shape (transforms.py:75-76); interp: t as is, y / x - 0.5
  interp_features = np.zeros((b, n, t, c), dt)
  for bi in range(b):
    for ti in range(t):
      pts = np.stack([np.full(n, ti, dt), tracks[bi, :, ti, 1] * dt.type(h / im_hw[0]),
                      tracks[bi, :, ti, 0] * dt.type(w / im_hw[1])], -1)
      interp_features[bi, :, ti] = interp_nearest_3d(feature_grid[bi], pts)
  query_frame = np.round(query_points[..., 0].astype(dt)).astype(np.int32)     # grid frames = video frames
  target = np.stack([fg[bi][np.clip(query_frame[bi], 0, t - 1)] for bi in range(b)])   # [B,N,h,w,C]
  dots2 = np.einsum('bntc,bnhwc->bnthw', rnd(interp_features), target, optimize=True).astype(dt)
  inverse = soft_points(dots2, None)
  dist = ((inverse - query_points[:, :, None, 2:0:-1].astype(dt)) ** 2).sum(-1)
  occlusion = (dist > dt.type(dist_threshold) ** 2).astype(dt) * dt.type(20.0) - dt.type(10.0)
  if return_stages:
    return tracks, occlusion, dict(inverse_tracks=inverse, interp_features=interp_features, dist=dist)
  return tracks, occlusion


# ---------------------------------------------------------------------------
# R3: PIPs patch correlation (front half of refine_pips)
# ---------------------------------------------------------------------------
def interp_constant_2d(grid, coords_yx):
  """model_utils.interp(mode='constant') per channel (model_utils.py:196-206).

  grid [h,w,C]; coords_yx [...,2] grid units (before the -0.5 shift).  Taps
  that fall outside the grid contribute zero (cval=0).  Returns [...,C].
  """
  dt = grid.dtype
  h, w, _ = grid.shape
  c = coords_yx.astype(dt) - dt.type(0.5)
  lo = np.floor(c)
  wu = (c - lo).astype(dt)
  wl = (dt.type(1) - wu).astype(dt)
  lo = lo.astype(np.int64)
  out = np.zeros(c.shape[:-1] + (grid.shape[-1],), dtype=dt)
  for dy in (0, 1):
    iy = lo[..., 0] + dy
    wy = wu[..., 0] if dy else wl[..., 0]
    vy = (iy >= 0) & (iy < h)
    for dx in (0, 1):
      ix = lo[..., 1] + dx
      wx = wu[..., 1] if dx else wl[..., 1]
      v = vy & (ix >= 0) & (ix < w)
      val = grid[np.clip(iy, 0, h - 1), np.clip(ix, 0, w - 1), :]
      out += val * (wy * wx * v.astype(dt))[..., None]
  return out


def patch_correlation(query, grid, pos_xy, orig_hw, last_iter_query=None):
  """One pyramid level of refine_pips (tapir_model.py:496-541), gather path.

  query [B,N,C]; grid [B,T,h,w,C]; pos_xy [B,N,T,2] in orig_hw coordinates;
  last_iter_query [B,N,T,C] or None.  Returns [B,N,T,49], s=(dy+3)*7+(dx+3).
  """
  dt = grid.dtype
  b, t, h, w, c = grid.shape
  orig_h, orig_w = orig_hw
  coords = convert_grid_coordinates(pos_xy.astype(dt), (orig_w, orig_h), (w, h))
  coords = coords[..., ::-1]  # (y, x)
  ctxx, ctxy = np.meshgrid(np.arange(-3, 4), np.arange(-3, 4))
  ctx = np.stack([ctxy, ctxx], axis=-1).reshape(-1, 2).astype(dt)  # dy outer
  coords2 = coords[:, :, :, None, :] + ctx[None, None, None]  # [B,N,T,49,2]
  n = coords.shape[1]
  out = np.zeros((b, n, t, 49), dtype=dt)
  for bi in range(b):
    for ti in range(t):
      nb = interp_constant_2d(grid[bi, ti], coords2[bi, :, ti])  # [N,49,C]
      if last_iter_query is None:
        out[bi, :, ti] = np.einsum('nsc,nc->ns', nb, query[bi])
      else:
        out[bi, :, ti] = np.einsum('nsc,nc->ns', nb, last_iter_query[bi, :, ti])
  return out


# ---------------------------------------------------------------------------
# R4: PIPs MLP-mixer
# ---------------------------------------------------------------------------
def layernorm(x, scale, eps=1e-5):
  """hk.LayerNorm(axis=-1, create_scale=True, create_offset=False) (tapir_model.py:33-36)."""
  dt = x.dtype
  mean = x.mean(axis=-1, keepdims=True, dtype=dt)
  var = np.mean(np.square(x - mean), axis=-1, keepdims=True, dtype=dt)
  return ((x - mean) / np.sqrt(var + dt.type(eps)) * scale.astype(dt)).astype(dt)


def gelu_tanh(x):
  """jax.nn.gelu default (approximate=True)."""
  dt = x.dtype
  k = dt.type(math.sqrt(2.0 / math.pi))
  return (dt.type(0.5) * x * (dt.type(1) + np.tanh(k * (x + dt.type(0.044715) * x * x * x)))).astype(dt)


def _depthwise_conv1d(x, w, b, mult, causal):
  """hk.DepthwiseConv1D over time, NWC (tapir_model.py:59-82).

  x [N,T,C]; w [C*mult,1,3] (torch layout, out channel = c*mult+m); b [C*mult].
  SAME (pad 1/1) or causal (pad 2/0).
  """
  dt = x.dtype
  n, t, c = x.shape
  pad = (2, 0) if causal else (1, 1)
  xp = np.pad(x, ((0, 0), pad, (0, 0)))
  xr = np.repeat(xp, mult, axis=-1) if mult > 1 else xp  # channel c*mult+m <- c
  out = np.zeros((n, t, c * mult), dtype=dt)
  for k in range(3):
    out += xr[:, k: k + t, :] * w[:, 0, k].astype(dt)
  return out + b.astype(dt)


def pips_conv_block(weights, prefix, x, use_causal_conv=False,
                    causal_context=None, get_causal_context=False,
                    block_name=None, rnd=None):
  """PIPsConvBlock (tapir_model.py:101-124) with depthwise_conv_residual (:39-89)."""
  rnd = rnd or _id
  dt = x.dtype
  to_skip = x
  x = layernorm(x, weights[prefix + 'layer_norm.weight'])
  new_ctx = {}
  num_extra = 0
  name1, name2 = block_name + '_causal_1', block_name + '_causal_2'
  if causal_context is not None:
    x = np.concatenate([causal_context[name1].astype(dt), x], axis=-2)
    num_extra = causal_context[name1].shape[-2]
  if get_causal_context:
    new_ctx[name1] = x[..., -2:, :].copy()
  x = _depthwise_conv1d(x, weights[prefix + 'mlp1_up.weight'],
                        weights[prefix + 'mlp1_up.bias'], 4, use_causal_conv)
  x = gelu_tanh(x)
  if causal_context is not None:
    x = np.concatenate([causal_context[name2].astype(dt), x[..., num_extra:, :]], axis=-2)
    num_extra = causal_context[name2].shape[-2]
  if get_causal_context:
    new_ctx[name2] = x[..., -2:, :].copy()
  x = _depthwise_conv1d(x, weights[prefix + 'mlp1_up_1.weight'],
                        weights[prefix + 'mlp1_up_1.bias'], 1, use_causal_conv)
  if causal_context is not None:
    x = x[..., num_extra:, :]
  x = x[..., 0::4] + x[..., 1::4] + x[..., 2::4] + x[..., 3::4]
  x = x + to_skip
  to_skip = x
  x = rnd(layernorm(x, weights[prefix + 'layer_norm_1.weight']))
  x = x @ rnd(weights[prefix + 'conv_channels_mixer.mlp2_up.weight']).T.astype(dt) \
      + weights[prefix + 'conv_channels_mixer.mlp2_up.bias'].astype(dt)
  x = rnd(gelu_tanh(x))
  x = x @ rnd(weights[prefix + 'conv_channels_mixer.mlp2_down.weight']).T.astype(dt) \
      + weights[prefix + 'conv_channels_mixer.mlp2_down.bias'].astype(dt)
  return (x + to_skip).astype(dt), new_ctx


def pips_mlp_mixer(weights, x, num_blocks=12, use_causal_conv=False,
                   causal_context=None, get_causal_context=False, rnd=None):
  """PIPSMLPMixer (tapir_model.py:127-156).  x [N,T,Cin] -> [N,T,388].

  Causal context keys follow the torch twin: ``block_{i}_causal_{1,2}``
  (tapnet/torch/tapir_model.py:766-768).
  """
  p = 'torch_pips_mixer.'
  rn = rnd or _id
  dt = x.dtype
  x = rn(x) @ rn(weights[p + 'linear.weight']).T.astype(dt) + weights[p + 'linear.bias'].astype(dt)
  all_ctx = {}
  for i in range(num_blocks):
    x, ctx = pips_conv_block(weights, f'{p}blocks.{i}.', x, use_causal_conv,
                             causal_context, get_causal_context,
                             block_name=f'block_{i}', rnd=rnd)
    all_ctx.update(ctx)
  x = rn(layernorm(x, weights[p + 'layer_norm.weight']))
  x = x @ rn(weights[p + 'linear_1.weight']).T.astype(dt) + weights[p + 'linear_1.bias'].astype(dt)
  return x.astype(dt), all_ctx


def refine_pips(weights, target_feature, pyramid, pos_guess, occ_guess, expd_guess,
                orig_hw, last_iter=None, resize_hw=None, num_blocks=12,
                use_causal_conv=False, causal_context=None,
                get_causal_context=False, return_stages=False, rnd=None):
  """TAPIR.refine_pips (tapir_model.py:473-624).

  target_feature: list of [B,N,C_l]; pyramid: list of [B,T,h_l,w_l,C_l];
  pos_guess [B,N,T,2] (x,y) in orig_hw px; last_iter [B,N,T,384] or None.
  causal_context: dict name -> [B,N,2,C].
  """
  dt = pyramid[0].dtype
  orig_h, orig_w = orig_hw
  resized_h, resized_w = resize_hw
  assert len(target_feature) == len(pyramid)
  corrs = []
  for pyridx, (query, grid) in enumerate(zip(target_feature, pyramid)):
    liq = None
    if last_iter is not None:
      liq = last_iter[..., :HIRES_DIM] if pyridx == 0 else last_iter[..., HIRES_DIM:]
    corrs.append(patch_correlation(query, (rnd or _id)(grid), pos_guess, orig_hw, liq))
  corrs = np.concatenate(corrs, axis=-1)
  b, n, t = pos_guess.shape[:3]
  if last_iter is None:
    both = np.concatenate([target_feature[0], target_feature[1]], axis=-1)
    feats = np.tile(both[:, :, None, :], (1, 1, t, 1))
  else:
    feats = last_iter
  mlp_input = np.concatenate([
      np.zeros_like(pos_guess), occ_guess[..., None], expd_guess[..., None],
      feats, corrs], axis=-1).astype(dt)
  x = mlp_input.reshape(b * n, t, -1)
  cc = None
  if causal_context is not None:
    cc = {k: v.reshape((b * n,) + v.shape[2:]) for k, v in causal_context.items()}
  res, new_cc = pips_mlp_mixer(weights, x, num_blocks, use_causal_conv, cc,
                               get_causal_context, rnd=rnd)
  res = res.reshape(b, n, t, -1)
  new_cc = {k: v.reshape((b, n) + v.shape[1:]) for k, v in new_cc.items()}
  pos_update = convert_grid_coordinates(res[..., :2], (resized_w, resized_h),
                                        (orig_w, orig_h))
  out = (pos_update + pos_guess, res[..., 2] + occ_guess,
         res[..., 3] + expd_guess, res[..., 4:] + feats, new_cc)
  if return_stages:
    return out + (dict(corrs=corrs, mlp_input=mlp_input, mixer_out=res),)
  return out


# ---------------------------------------------------------------------------
# R1 / R0: estimate_trajectories and the final averaging of __call__
# ---------------------------------------------------------------------------
def estimate_trajectories(weights, video_size, lowres, hires, resolutions,
                          q_lowres, q_hires, query_points_in_video,
                          num_pips_iter=4, pyramid_level=1,
                          softmax_temperature=20.0, initial_resolution=(256, 256),
                          num_blocks=12, use_causal_conv=False,
                          query_chunk_size=None, causal_context=None,
                          get_causal_context=False, rnd=None):
  """TAPIR.estimate_trajectories (tapir_model.py:858-1066).

  The random query permutation (:938-946) only randomises which chunk a query
  lands in; per-query results do not depend on it (tapnet/tapvid/README.md:32-38),
  so the oracle uses the identity permutation.
  Returns dict(occlusion, tracks, expected_dist [, causal_context]) of lists.
  """
  dt = lowres[0].dtype
  num_iters = num_pips_iter * (len(lowres) - 1)
  nq = q_lowres[0].shape[1]
  chunk = nq if not query_chunk_size else query_chunk_size
  occ_it = [[] for _ in range(num_iters + 1)]
  pts_it = [[] for _ in range(num_iters + 1)]
  exp_it = [[] for _ in range(num_iters + 1)]
  cc_it = [[] for _ in range(num_iters)]
  num_frames = lowres[0].shape[1]

  def train2orig(x):
    return convert_grid_coordinates(x, initial_resolution[::-1], video_size[::-1])

  for ch in range(0, nq, chunk):
    sl = slice(ch, ch + chunk)
    qp = None
    if query_points_in_video is not None:
      qp = convert_grid_coordinates(
          query_points_in_video[:, sl].astype(dt),
          (num_frames,) + tuple(video_size),
          (num_frames,) + tuple(initial_resolution))
    points, occ, expd = tracks_from_cost_volume(
        weights, q_lowres[0][:, sl], lowres[0], qp, initial_resolution,
        softmax_temperature, rnd=rnd)
    pts_it[0].append(train2orig(points))
    occ_it[0].append(occ)
    exp_it[0].append(expd)
    mixer_feats = None
    for i in range(num_iters):
      lvl = i // num_pips_iter + 1
      queries = [q_hires[lvl][:, sl], q_lowres[lvl][:, sl]]
      pyramid = [hires[lvl], lowres[lvl]]
      for _ in range(pyramid_level):
        queries.append(queries[-1])
        pyramid.append(avg_pool_2x2(pyramid[-1]))
      cc = None
      if causal_context is not None:
        cc = {k: v[:, sl] for k, v in causal_context[i].items()}
      points, occ, expd, mixer_feats, new_cc = refine_pips(
          weights, queries, pyramid, points, occ, expd, initial_resolution,
          last_iter=mixer_feats, resize_hw=resolutions[lvl], num_blocks=num_blocks,
          use_causal_conv=use_causal_conv, causal_context=cc,
          get_causal_context=get_causal_context, rnd=rnd)
      pts_it[i + 1].append(train2orig(points))
      occ_it[i + 1].append(occ)
      exp_it[i + 1].append(expd)
      cc_it[i].append(new_cc)
      if (i + 1) % num_pips_iter == 0:
        mixer_feats = None
        expd = exp_it[0][-1]
        occ = occ_it[0][-1]
  out = dict(
      occlusion=[np.concatenate(v, axis=1) for v in occ_it],
      tracks=[np.concatenate(v, axis=1) for v in pts_it],
      expected_dist=[np.concatenate(v, axis=1) for v in exp_it])
  if get_causal_context:
    out['causal_context'] = [
        {k: np.concatenate([d[k] for d in lst], axis=1) for k in lst[0]}
        for lst in cc_it]
  return out


def tapir_from_grids(weights, video_shape, lowres, hires, resolutions, query_points,
                     num_pips_iter=4, **kw):
  """TAPIR.__call__ minus the backbone (tapir_model.py:1068-1154)."""
  ql, qh = get_query_features(lowres, hires, resolutions, query_points, video_shape)
  traj = estimate_trajectories(weights, tuple(video_shape[2:4]), lowres, hires,
                               resolutions, ql, qh, query_points,
                               num_pips_iter=num_pips_iter, **kw)
  p = num_pips_iter
  return dict(
      occlusion=np.mean(np.stack(traj['occlusion'][p::p]), axis=0),
      tracks=np.mean(np.stack(traj['tracks'][p::p]), axis=0),
      expected_dist=np.mean(np.stack(traj['expected_dist'][p::p]), axis=0),
      unrefined_occlusion=traj['occlusion'][:-1],
      unrefined_tracks=traj['tracks'][:-1],
      unrefined_expected_dist=traj['expected_dist'][:-1])


def construct_initial_causal_state(num_points, num_resolutions=1, num_blocks=12,
                                   dtype=np.float32):
  """TAPIR.construct_initial_causal_state (tapir_model.py:1156-1170; torch :763-772)."""
  ret = {}
  for i in range(num_blocks):
    ret[f'block_{i}_causal_1'] = np.zeros((1, num_points, 2, 512), dtype)
    ret[f'block_{i}_causal_2'] = np.zeros((1, num_points, 2, 2048), dtype)
  return [dict((k, v.copy()) for k, v in ret.items())
          for _ in range(num_resolutions * 4)]


def update_query_features(q_lowres, q_hires, new_q_lowres, new_q_hires, idx_to_update,
                          causal_state=None, num_blocks=12):
  """TAPIR.update_query_features (tapir_model.py:1172-1203; torch twin :774-806), functional
  like the JAX model: entries ``idx_to_update`` of every level's query features are replaced by
  the new ones and, if a causal state is given, the same entries of every state tensor are
  replaced by a fresh (zero) state (construct_initial_causal_state for len(idx) points).

  q_lowres / q_hires: per-level lists of [B,N,C]; new_*: per-level lists of [B,len(idx),C].
  Returns (q_lowres, q_hires) or (q_lowres, q_hires, causal_state)."""
  if isinstance(idx_to_update, int):
    idx_to_update = (idx_to_update,)
  idx = np.array(idx_to_update)

  def upd(s1, s2):
    out = np.array(s1, copy=True)
    out[:, idx] = s2
    return out

  ql = [upd(a, b) for a, b in zip(q_lowres, new_q_lowres)]
  qh = [upd(a, b) for a, b in zip(q_hires, new_q_hires)]
  if causal_state is None:
    return ql, qh
  init = construct_initial_causal_state(len(idx), len(q_lowres) - 1, num_blocks)
  assert len(init) == len(causal_state)
  new_state = [{k: upd(d[k], z[k]) for k in d} for d, z in zip(causal_state, init)]
  return ql, qh, new_state


def cast_weights(weights, dtype):
  return {k: np.asarray(v).astype(dtype) for k, v in weights.items()}
