"""Bulk online tracking of many sampled points through many videos: the workload of
``tapnet/robotap/tapir_clustering.py:918-1179`` (`track_many_points` and its helpers
`query_features_join` :969-980, `query_features_count` :983-985,
`predictions_to_tracks_visibility` :988-1012, `preprocess_frames` :1015-1026,
`construct_fake_causal_state` :834-853) on the MI355X engine.

What the reference does: sample `points_per_frame` random points on every `frame_stride`-th frame
of every video, extract their query features (one model call per sampled frame), group them in
batches of `point_batch_size` points (the last batch padded by repeating its last frame's points),
and for every batch stream EVERY video frame by frame through the causal model from a zero state,
keeping the last refinement iteration's tracks and the thresholded visibility.

How it runs here (same results, different schedule):
  * query features of all sampled frames of a video come from ONE backbone pass over those frames
    (InstanceNorm is per frame and a query's features are sampled from its own frame only, so the
    batch gives what the per-frame calls give) and one `tapir_get_query_features` launch;
  * the per-frame step of a point batch is an `OnlineTracker` session: backbone on the frame, cost
    volume, refinement iterations and the causal-state hand-over replayed from one captured
    hipGraph per direction, with the state ping-ponging between two packed device buffers -- no
    per-frame allocation, no host synchronisation except the copy-out of the frame's results;
  * results of a whole video stay on the GPU ([N, T, 2] and [N, T]) and are copied out once.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from tapnet_amd.tapir_model import TAPIR, QueryFeatures


def query_features_join(feature_list: Sequence[QueryFeatures]) -> QueryFeatures:
  """tapir_clustering.py:969-980: concatenate along the point axis."""
  cat = (lambda xs: torch.cat(list(xs), dim=1)) if torch.is_tensor(feature_list[0].lowres[0]) else \
      (lambda xs: np.concatenate(list(xs), axis=1))
  return QueryFeatures(
      lowres=tuple(cat(x) for x in zip(*[f.lowres for f in feature_list])),
      hires=tuple(cat(x) for x in zip(*[f.hires for f in feature_list])),
      resolutions=feature_list[0].resolutions)


def query_features_count(features: QueryFeatures) -> int:
  """tapir_clustering.py:983-985."""
  return int(features.lowres[0].shape[1])


def query_features_slice(features: QueryFeatures, lo: int, hi: int) -> QueryFeatures:
  return QueryFeatures(tuple(x[:, lo:hi] for x in features.lowres),
                       tuple(x[:, lo:hi] for x in features.hires), features.resolutions)


def _sigmoid(x):
  return torch.sigmoid(x) if torch.is_tensor(x) else 0.5 * (1.0 + np.tanh(0.5 * np.asarray(x, np.float32)))


def predictions_to_tracks_visibility(predictions: Mapping[str, Any], single_step: bool = True):
  """tapir_clustering.py:988-1012: tracks [N,(T),2] and visibility in [0,1] [N,(T)] =
  (1 - sigmoid(occlusion)) * (1 - sigmoid(expected_dist)) of batch element 0."""
  tracks = predictions['tracks'][0]
  occlusion = predictions['occlusion'][0]
  expected_dist = predictions['expected_dist'][0]
  if single_step:
    tracks, occlusion, expected_dist = tracks[:, 0], occlusion[:, 0], expected_dist[:, 0]
  visibility = (1 - _sigmoid(occlusion)) * (1 - _sigmoid(expected_dist))
  return tracks, visibility


def preprocess_frames(frames):
  """tapir_clustering.py:1015-1026: uint8 [0,255] -> float32 [-1,1]."""
  if torch.is_tensor(frames):
    return frames.to(torch.float32) / 255 * 2 - 1
  return np.asarray(frames).astype(np.float32) / 255 * 2 - 1


def sample_query_points(video_shapes: Sequence[Tuple[int, ...]], frame_stride: int, points_per_frame: int,
                        sample_box_corners=(0.1, 0.1, 0.9, 0.9), seed: int = 42):
  """The sampling of tapir_clustering.py:1073-1090, in its draw order (``np.random.seed(42)``, one
  uniform [points_per_frame, 3] draw per sampled frame, videos outer, frames inner).  Returns a list
  of (video_index, frame_index, yx [P,2]) per sampled frame."""
  rng = np.random.RandomState(seed)
  x_scl = sample_box_corners[2] - sample_box_corners[0]
  y_scl = sample_box_corners[3] - sample_box_corners[1]
  x_add, y_add = sample_box_corners[0], sample_box_corners[1]
  out = []
  for v, shp in enumerate(video_shapes):
    t, h, w = shp[0], shp[1], shp[2]
    for i in range(0, t, frame_stride):
      qp = (rng.uniform(0.0, 1.0, [points_per_frame, 3]) * np.array([0.0, h * y_scl, w * x_scl])[None]
            + np.array([0.0, h * y_add, w * x_add])[None])
      out.append((v, i, qp[:, 1:]))
  return out


def extract_query_features(model: TAPIR, video_u8, frame_indices: Sequence[int],
                           yx_per_frame: Sequence[np.ndarray], max_frames_per_pass: int = 64) -> QueryFeatures:
  """Query features of points given per sampled frame (the reference calls its init model once per
  sampled frame with that single frame, tapir_clustering.py:1092-1096).  Here: one backbone pass per
  `max_frames_per_pass` sampled frames and one query-feature launch; the query's time coordinate is
  the index of its frame inside the pass, an integer, so the trilinear sample reads that frame only."""
  parts = []
  for lo in range(0, len(frame_indices), max_frames_per_pass):
    idx = list(frame_indices[lo:lo + max_frames_per_pass])
    frames = preprocess_frames(torch.as_tensor(np.asarray(video_u8)[idx]))[None]      # [1,F,H,W,3]
    qp = np.concatenate([np.concatenate([np.full((len(yx), 1), f, np.float64), yx], axis=1)
                         for f, yx in enumerate(yx_per_frame[lo:lo + max_frames_per_pass])], axis=0)
    frames = model._dev(frames)
    fg = model.get_feature_grids(frames)
    qf = model.get_query_features(frames, False, model._dev(torch.as_tensor(qp[None], dtype=torch.float32)), fg)
    parts.append(QueryFeatures(tuple(t.clone() for t in qf.lowres), tuple(t.clone() for t in qf.hires),
                               qf.resolutions))
  return query_features_join(parts)


def track_points_in_video(tracker, query_features: QueryFeatures, video_u8, visibility_threshold: float = 0.5):
  """One point batch through one video from a zero causal state (tapir_clustering.py:1133-1151):
  tracks [N,T,2] float32 (x,y) and visibility [N,T] bool, as device tensors."""
  tracker.set_query_features(query_features)
  video_u8 = torch.as_tensor(np.asarray(video_u8))
  T = video_u8.shape[0]
  dev = tracker.model.device
  n = query_features_count(query_features)
  tracks = torch.empty((n, T, 2), device=dev)
  visible = torch.empty((n, T), device=dev, dtype=torch.bool)
  frames = preprocess_frames(video_u8.to(dev, non_blocking=True))
  for t in range(T):
    pred = tracker.step(frames[t])
    trk, vis = predictions_to_tracks_visibility(pred)
    tracks[:, t] = trk
    visible[:, t] = vis > visibility_threshold
  return tracks, visible


def track_many_points(separation_videos: Mapping[Any, np.ndarray], demo_episode_ids: Sequence[Any],
                      model: TAPIR, frame_stride: int = 4, points_per_frame: int = 8,
                      point_batch_size: int = 2048, sample_box_corners=(0.1, 0.1, 0.9, 0.9),
                      seed: int = 42, tracker_factory: Optional[Callable[..., Any]] = None) -> Dict[str, Any]:
  """tapir_clustering.py:1029-1179 with `checkpoint_path` replaced by a constructed causal `model`
  (``TAPIR(use_causal_conv=True, ...)``).  Same return structure: `separation_tracks` /
  `separation_visibility` are dicts episode id -> [num_points, T_episode, 2] / [num_points, T_episode]
  holding, for EVERY sampled point (from any video), its track through that episode."""
  if not model.use_causal_conv:
    raise ValueError('Online model requires causal TAPIR training.')
  if point_batch_size % points_per_frame:
    raise ValueError('point_batch_size must be a multiple of points_per_frame')
  if tracker_factory is None:
    from tapnet_amd.online import OnlineTracker
    tracker_factory = OnlineTracker
  videos = [np.asarray(separation_videos[x]) for x in demo_episode_ids]
  shapes = [v.shape for v in videos]
  if len({s[1:3] for s in shapes}) != 1:
    raise ValueError('all videos must share one frame size (one captured step per size)')
  samples = sample_query_points(shapes, frame_stride, points_per_frame, sample_box_corners, seed)

  # -- query features, video by video
  feats, q_video, q_frame, q_yx = [], [], [], []
  for v, video in enumerate(videos):
    mine = [(i, yx) for (vv, i, yx) in samples if vv == v]
    if not mine:
      continue
    feats.append(extract_query_features(model, video, [i for i, _ in mine], [yx for _, yx in mine]))
    for i, yx in mine:
      q_video.append(np.full(points_per_frame, v)); q_frame.append(np.full(points_per_frame, i)); q_yx.append(yx)
  all_features = query_features_join(feats)
  n_points = query_features_count(all_features)
  out_query_points = [np.concatenate(q_video), np.concatenate(q_frame), np.concatenate(q_yx, axis=0)]

  # -- batches of point_batch_size; the last one is padded by repeating its last frame's points
  #    (tapir_clustering.py:1115-1123) so that every batch runs the same captured step
  tracker = tracker_factory(model, point_batch_size, shapes[0][1:3])
  all_tracks, all_vis = [], []
  for lo in range(0, n_points, point_batch_size):
    hi = min(lo + point_batch_size, n_points)
    batch = query_features_slice(all_features, lo, hi)
    num_extra = point_batch_size - (hi - lo)
    if num_extra:
      last = query_features_slice(all_features, hi - points_per_frame, hi)
      batch = query_features_join([batch] + [last] * (num_extra // points_per_frame))
    b_tracks, b_vis = [], []
    for video in videos:
      trk, vis = track_points_in_video(tracker, batch, video)
      b_tracks.append(trk[:hi - lo]); b_vis.append(vis[:hi - lo])
    all_tracks.append(torch.cat(b_tracks, dim=1)); all_vis.append(torch.cat(b_vis, dim=1))
  tracks = torch.cat(all_tracks, dim=0).cpu().numpy()
  visibility = torch.cat(all_vis, dim=0).cpu().numpy()

  bnds, cur = [], 0
  for shp in shapes:
    bnds.append((cur, cur + shp[0])); cur += shp[0]
  to_np = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
  return {
      'separation_visibility': {k: visibility[:, lb:ub] for k, (lb, ub) in zip(demo_episode_ids, bnds)},
      'separation_tracks': {k: tracks[:, lb:ub] for k, (lb, ub) in zip(demo_episode_ids, bnds)},
      'video_shape': {x: shapes[i] for i, x in enumerate(demo_episode_ids)},
      'query_features': QueryFeatures(tuple(to_np(t) for t in all_features.lowres),
                                      tuple(to_np(t) for t in all_features.hires), all_features.resolutions),
      'demo_episode_ids': demo_episode_ids,
      'query_points': out_query_points,
  }
