"""Drop-in mirror of ``tapnet.models.tapir_model`` (the reference's API for the
TAPIR inference path) on top of libtapir_hip.so.

Same class / method names, argument meaning, output dict keys and error
behaviour as the reference (tapnet/models/tapir_model.py):

    TAPIR(...).__call__ / get_feature_grids / get_query_features /
    estimate_trajectories / tracks_from_cost_volume / refine_pips /
    construct_initial_causal_state / update_query_features,
    FeatureGrids, QueryFeatures, ParameterizedTAPIR, and the north_star alias
    build_cost_volume (= the einsum of tracks_from_cost_volume :433).

Everything after the feature grids runs in hand-written gfx950 kernels through
the C ABI (include/tapir_hip.h); PyTorch-ROCm is used for the ResNet backbone,
for device memory and streams.  There is no CPU fallback: a missing library or
a non-GPU tensor device raises.

Differences from the JAX signature, all additive:
  * weights are passed to the constructor (``weights=`` flat dict keyed by the
    reference's torch state_dict names, e.g. a loaded ``.pt`` checkpoint);
    ``ParameterizedTAPIR(params, state, tapir_kwargs)`` keeps the Haiku-style
    constructor and accepts either that dict or Haiku params.
  * ``dtype='float32'|'bfloat16'`` selects the arithmetic of the GEMM-shaped
    stages (exact-f32 MFMA parity build vs bf16 MFMA speed build).
  * inputs may be numpy arrays or torch tensors; numpy in -> numpy out.
  * ``query_chunk_size`` is accepted and ignored: the fused kernels never
    materialise the [T,B,N,h,w] volume the chunking exists to bound
    (tapir_model.py:880-881), and per-query results do not depend on it.
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, List, Mapping, NamedTuple, Optional, Sequence, Tuple

import numpy as np
import torch

from tapnet_amd import _ffi
from tapnet_amd import backbone as backbone_lib
from tapnet_amd import model_utils
from tapnet_amd.model_utils import convert_grid_coordinates

HIRES_DIM = 128
LOWRES_DIM = 256


class FeatureGrids(NamedTuple):
  """tapir_model.py:251-270.  ``resolutions`` are (H, W) tuples (torch twin convention) that also carry
  ``.shape`` (JAX convention): see Resolution."""
  lowres: Sequence[Any]
  hires: Sequence[Any]
  resolutions: Sequence[Tuple[int, int]]


class StagedFeatureGrids(FeatureGrids):
  """FeatureGrids that also carry, for the bf16 engine, the hot path's operand-type copies of their grids:
  ``staged`` = [(f32 grid, bf16 row-major copy, bf16 tile-order copy or None), ...].  TAPIR.__call__ registers them
  with the engine for exactly that call (tapir_set_staged_grid) instead of re-casting the f32 grids.  Produced by
  tapnet_amd.distributed.gather_feature_grids, which gathers the bf16 copies the backbone's L2-normalise kernel wrote
  (they ARE the wire format); a plain FeatureGrids works everywhere this one does."""
  staged: Any = None


class QueryFeatures(NamedTuple):
  """tapir_model.py:273-293."""
  lowres: Sequence[Any]
  hires: Sequence[Any]
  resolutions: Sequence[Tuple[int, int]]


class CausalState(list):
  """List (one entry per refinement iteration) of dicts name -> [B,N,2,C] like the
  reference's causal context (tapir_model.py:1156-1170).  The dict values are views
  of two packed device tensors the kernels use directly."""
  packed: Optional[Tuple[torch.Tensor, torch.Tensor]] = None


def _is_numpy(x) -> bool:
  return isinstance(x, np.ndarray)


class Resolution(tuple):
  """An (H, W) tuple -- the torch twin's convention (tapnet/torch/tapir_model.py:45) -- that also answers
  ``.shape`` like the JAX model's zero-size shape carriers (``video_resize[0, 0, :, :, 0:0]``, :259-265, :724),
  so ``feature_grids.resolutions[i].shape[:2]`` and ``resolutions[i][0]`` both read the resolution."""

  @property
  def shape(self) -> Tuple[int, int, int]:
    return (int(self[0]), int(self[1]), 0)


def _res_hw(r) -> Tuple[int, int]:
  """Accepts (H, W) tuples or the JAX model's zero-size shape carriers (:259-265)."""
  if hasattr(r, 'shape') and not isinstance(r, (tuple, list)):
    return int(r.shape[0]), int(r.shape[1])
  return int(r[0]), int(r[1])


def causal_block_names(num_blocks: int, haiku: bool = False) -> List[Tuple[str, str]]:
  """Keys of the causal context: torch twin ``block_{i}_causal_{1,2}``
  (tapnet/torch/tapir_model.py:766-768) or the Haiku names (:1159-1166)."""
  out = []
  for i in range(num_blocks):
    if haiku:
      bid = '' if i == 0 else f'{i}_'
      base = f'tapir/~/pips_mlp_mixer/block_{bid}causal'
    else:
      base = f'block_{i}_causal'
    out.append((base + '_1', base + '_2'))
  return out


class TAPIR:
  """TAPIR model (tapir_model.py:296-397) -- MI355X engine."""

  def __init__(
      self,
      bilinear_interp_with_depthwise_conv: bool = False,
      num_pips_iter: int = 4,
      pyramid_level: int = 1,
      mixer_hidden_dim: int = 512,
      num_mixer_blocks: int = 12,
      mixer_kernel_shape: int = 3,
      patch_size: int = 7,
      softmax_temperature: float = 20.0,
      use_causal_conv: bool = False,
      parallelize_query_extraction: bool = False,
      initial_resolution: Tuple[int, int] = (256, 256),
      blocks_per_group: Sequence[int] = (2, 2, 2, 2),
      extra_convs: bool = False,
      extra_convs_kwargs=None,
      feature_extractor_chunk_size: Optional[int] = None,
      name: str = 'tapir',
      *,
      weights: Optional[Mapping[str, Any]] = None,
      dtype: str = 'float32',
      device: Any = None,
      haiku_state_names: bool = False,
      use_casual_conv: Optional[bool] = None,
      jax_antialias_resize: bool = False,
  ):
    del bilinear_interp_with_depthwise_conv, parallelize_query_extraction, name
    if use_casual_conv is not None:   # the torch twin spells the argument this way (torch/tapir_model.py:84)
      use_causal_conv = use_casual_conv
    if mixer_hidden_dim != 512 or mixer_kernel_shape != 3 or patch_size != 7:
      raise ValueError('the HIP kernels are built for mixer_hidden_dim=512, '
                       'mixer_kernel_shape=3, patch_size=7 (the released checkpoints)')
    if extra_convs_kwargs:
      raise ValueError('extra_convs_kwargs other than the defaults are not supported')
    if dtype not in ('float32', 'bfloat16'):
      raise ValueError("dtype must be 'float32' or 'bfloat16'")
    self.num_pips_iter = num_pips_iter
    self.pyramid_level = pyramid_level
    self.num_mixer_blocks = num_mixer_blocks
    self.softmax_temperature = float(softmax_temperature)
    self.use_causal_conv = bool(use_causal_conv)
    self.initial_resolution = tuple(initial_resolution)
    self.blocks_per_group = tuple(blocks_per_group)
    self.extra_convs = bool(extra_convs)
    self.feature_extractor_chunk_size = feature_extractor_chunk_size
    self.highres_dim, self.lowres_dim = HIRES_DIM, LOWRES_DIM
    self.dtype = dtype
    self.haiku_state_names = haiku_state_names
    # jax.image.resize anti-aliases when down-sampling (tapir_model.py:670), the torch twin does not
    # (tapnet/torch/utils.py:39): False = the torch twin (pinned by the fixtures), True = the JAX text
    self.jax_antialias_resize = bool(jax_antialias_resize)

    if device is None:
      device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type != 'cuda':
      raise RuntimeError('tapnet_amd.TAPIR needs a ROCm GPU (torch device "cuda"); '
                         'there is no CPU path')
    self.device = torch.device(device)
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())

    self._lib = _ffi.load_library()
    self._staged = []   # (f32 grid, bf16 copy, tile-order copy) of the grids borrowed by the running __call__
    self._ctx = ctypes.c_void_p()
    cfg = _ffi.TapirCfg(pyramid_level, num_pips_iter, num_mixer_blocks, int(use_causal_conv),
                        self.softmax_temperature, self.initial_resolution[0],
                        self.initial_resolution[1],
                        _ffi.TAPIR_BF16 if dtype == 'bfloat16' else _ffi.TAPIR_F32)
    rc = self._lib.tapir_create(ctypes.byref(self._ctx), ctypes.byref(cfg), self.device.index)
    if rc != _ffi.TAPIR_OK:
      raise ValueError(f'tapir_create failed ({rc}): unsupported configuration')
    self._backbone = None
    self._weights_loaded = False
    if weights is not None:
      self.load_weights(weights)

  # ------------------------------------------------------------------ plumbing
  def __del__(self):
    try:
      if getattr(self, '_backbone', None) is not None:
        self._backbone._graphs = {}
      if getattr(self, '_ctx', None):
        self._lib.tapir_destroy(self._ctx)
        self._ctx = None
    except Exception:  # interpreter shutdown
      pass

  def _check(self, rc, what):
    _ffi.check(self._lib, self._ctx, rc, what)

  def _stream(self):
    return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def _dev(self, x, dtype=torch.float32) -> torch.Tensor:
    t = torch.as_tensor(x) if not isinstance(x, torch.Tensor) else x
    return t.to(device=self.device, dtype=dtype).contiguous()

  @staticmethod
  def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())

  def load_weights(self, weights: Mapping[str, Any]):
    """weights: flat dict keyed by the reference's torch state_dict names
    (tapnet/torch/tapir_model.py:115-137), numpy arrays or torch tensors."""
    host = {}
    for k, v in weights.items():
      a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
      host[k] = np.ascontiguousarray(a, dtype=np.float32)
    for k, a in host.items():
      if k.startswith('resnet_torch.') or k.startswith('extra_convs.'):
        continue
      shape = (ctypes.c_int64 * a.ndim)(*a.shape)
      self._check(self._lib.tapir_set_weight(self._ctx, k.encode(), a.ctypes.data_as(ctypes.c_void_p),
                                             shape, a.ndim), f'tapir_set_weight({k})')
    self._check(self._lib.tapir_finalize_weights(self._ctx), 'tapir_finalize_weights')
    # (a dict with hot-path weights only keeps the current backbone: its packed weight streams belong to
    # the Backbone object, not to the hot-path weights tapir_finalize_weights rebuilds)
    if any(k.startswith('resnet_torch.') for k in host):
      if self._backbone is not None:
        torch.cuda.synchronize(self.device)
        self._backbone.close()
      self._backbone = backbone_lib.Backbone(
          host, self.extra_convs, self.device,
          torch.bfloat16 if self.dtype == 'bfloat16' else torch.float32, self.blocks_per_group,
          engine=(self._lib, self._ctx))
    self._weights_loaded = True

  # -- the call surface of the reference's PyTorch twin (tapnet/torch/tapir_model.py:139-147 and its
  #    use in tapnet/pytorch_live_demo.py:110-116): same model, nn.Module-style entry points
  def load_state_dict(self, state_dict: Mapping[str, Any], strict: bool = True):
    """``model.load_state_dict(torch.load(checkpoint.pt))`` (pytorch_live_demo.py:111-113)."""
    del strict
    self.load_weights(state_dict)
    return self

  def to(self, *args, **kwargs):
    """The engine lives on the GPU it was created on; moving it is a no-op (returns self)."""
    del args, kwargs
    return self

  def eval(self):
    return self

  def train(self, mode: bool = True):
    if mode:
      raise ValueError('tapnet_amd.TAPIR is an inference engine: training mode is not supported')
    return self

  def forward(self, video, query_points, is_training: bool = False, query_chunk_size: Optional[int] = 64,
              get_query_feats: bool = False,
              refinement_resolutions: Optional[List[Tuple[int, int]]] = None):
    """torch/tapir_model.py:139-215: the torch twin's argument order (query_points second)."""
    return self(video, is_training, query_points, query_chunk_size=query_chunk_size,
                get_query_feats=get_query_feats, refinement_resolutions=refinement_resolutions)

  def reserve(self, batch: int, num_queries: int, num_frames: int, lowres_hw: Tuple[int, int]):
    """Pre-sizes all workspaces (needed before hipGraph capture)."""
    self._check(self._lib.tapir_reserve(self._ctx, batch, num_queries, num_frames,
                                        lowres_hw[0], lowres_hw[1]), 'tapir_reserve')

  def profile_enable(self, on=True):
    """hipEvent brackets around the hot kernels (include/tapir_hip.h, measurement support).
    on: True = every kernel class, False = off, or an iterable of class names (_ffi.PROF_KINDS)."""
    if on is True:
      mask = -1
    elif not on:
      mask = 0
    else:
      mask = 0
      for name in on:
        mask |= 1 << _ffi.PROF_KINDS[name]
    self._check(self._lib.tapir_profile_enable(self._ctx, mask), 'tapir_profile_enable')

  def profile_stride(self, stride: int = 1) -> None:
    """Only every stride-th launch of an enabled class carries events (a timed launch costs ~12 us of idle device on either
    side of it); profile_read() then returns the sum and the count of the SAMPLED launches."""
    self._check(self._lib.tapir_profile_stride(self._ctx, int(stride)), 'tapir_profile_stride')

  def profile_read(self) -> Dict[str, Tuple[float, int]]:
    """{kernel class: (summed ms, launches)} since the last read; synchronises on the events."""
    out = {}
    for name, kind in _ffi.PROF_KINDS.items():
      ms, n = ctypes.c_double(), ctypes.c_int64()
      self._check(self._lib.tapir_profile_read(self._ctx, kind, ctypes.byref(ms), ctypes.byref(n)),
                  'tapir_profile_read')
      out[name] = (ms.value, n.value)
    return out

  # ------------------------------------------------------------------ R7
  def get_feature_grids(self, video, is_training: bool = False,
                        refinement_resolutions: Optional[List[Tuple[int, int]]] = None,
                        _borrow: bool = False, _global_frames: Optional[int] = None) -> FeatureGrids:
    """tapir_model.py:626-729.  (_borrow: internal -- __call__ consumes the grids before it returns
    and lets the backbone hand out its graph's own output buffers.  _global_frames: internal --
    frames of the whole clip when `video` is one rank's shard, tapnet_amd.distributed.)"""
    del is_training
    if self._backbone is None:
      raise RuntimeError('backbone weights (resnet_torch.*) were not loaded')
    video = self._dev(video)
    if refinement_resolutions is None:
      refinement_resolutions = model_utils.generate_default_resolutions(
          video.shape[2:4], self.initial_resolution)
    all_res = [tuple(self.initial_resolution)] + [tuple(r) for r in refinement_resolutions]
    feature_grid, hires_feats, resize_im_shape = [], [], []
    curr = (-1, -1)
    latent = hires = None
    for resolution in all_res:
      if resolution[0] % 8 != 0 or resolution[1] % 8 != 0:
        raise ValueError('Image resolution must be a multiple of 8.')
      if not model_utils.is_same_res(curr, resolution):
        # (quirk kept from the reference :667: compares the PREVIOUS resolution)
        if model_utils.is_same_res(curr, video.shape[-3:-1]):
          video_resize = video
        else:
          video_resize = backbone_lib.resize_bilinear(video, resolution, self.jax_antialias_resize)
        curr = resolution
        b, t, h, w, c = video_resize.shape
        low, hi = self._backbone.features(video_resize.reshape(b * t, h, w, c),
                                          self.feature_extractor_chunk_size, borrow=_borrow,
                                          global_frames=None if _global_frames is None else b * _global_frames)
        if _borrow and self._backbone.last_staged is not None:
          # bf16 engine: the backbone's L2-normalise kernel also wrote the hot path's operand-type copies
          low16, low_tiled, hi16 = self._backbone.last_staged
          self._staged.append((low, low16, low_tiled))
          self._staged.append((hi, hi16, None))
        latent = low.reshape(b, t, *low.shape[1:])
        hires = hi.reshape(b, t, *hi.shape[1:])
      feature_grid.append(latent)
      hires_feats.append(hires)
      resize_im_shape.append(Resolution(resolution))
    return FeatureGrids(tuple(feature_grid), tuple(hires_feats), tuple(resize_im_shape))

  # ------------------------------------------------------------------ R8
  def get_query_features(self, video, is_training: bool = False, query_points=None,
                         feature_grids: Optional[FeatureGrids] = None,
                         refinement_resolutions: Optional[List[Tuple[int, int]]] = None
                         ) -> QueryFeatures:
    """tapir_model.py:731-856.  ``video`` is only used for its shape (it may be a
    shape tuple when feature_grids is given)."""
    if feature_grids is None:
      feature_grids = self.get_feature_grids(video, is_training, refinement_resolutions)
    shape = tuple(video.shape) if hasattr(video, 'shape') else tuple(video)
    qp = self._dev(query_points)
    B, Q = qp.shape[:2]
    resolutions = [Resolution(_res_hw(r)) for r in feature_grids.resolutions]
    q_low, q_hi = [], []
    curr = (-1, -1)
    for i, res in enumerate(resolutions):
      if model_utils.is_same_res(curr, res):
        q_low.append(q_low[-1]); q_hi.append(q_hi[-1])
        continue
      curr = res
      outs = []
      for grid in (feature_grids.lowres[i], feature_grids.hires[i]):
        g = self._dev(grid)
        _, T, h, w, C = g.shape
        if shape[1] != T:
          raise ValueError('converting frame count is not supported.')
        out = torch.empty((B, Q, C), device=self.device, dtype=torch.float32)
        self._check(self._lib.tapir_get_query_features(
            self._ctx, self._ptr(g), self._ptr(qp), B, Q, T, h, w, C, int(shape[2]), int(shape[3]),
            self._ptr(out), self._stream()), 'tapir_get_query_features')
        outs.append(out)
      q_low.append(outs[0]); q_hi.append(outs[1])
    return QueryFeatures(tuple(q_low), tuple(q_hi), tuple(resolutions))

  # ------------------------------------------------------------------ R2
  def build_cost_volume(self, interp_feature, feature_grid):
    """einsum('bnc,bthwc->tbnhw') (tapir_model.py:433).  Returns [T,B,N,h,w]."""
    numpy_out = _is_numpy(interp_feature)
    qf, g = self._dev(interp_feature), self._dev(feature_grid)
    B, Q, C = qf.shape
    _, T, h, w, _ = g.shape
    vol = torch.empty((B, Q, T, h, w), device=self.device, dtype=torch.float32)
    self._check(self._lib.tapir_build_cost_volume(self._ctx, self._ptr(qf), self._ptr(g), B, Q, T,
                                                  h, w, C, self._ptr(vol), self._stream()),
                'tapir_build_cost_volume')
    vol = vol.permute(2, 0, 1, 3, 4)
    return vol.cpu().numpy() if numpy_out else vol

  def tracks_from_cost_volume(self, interp_feature, feature_grid, query_points, im_shp=None):
    """tapir_model.py:399-471.  query_points [B,N,3] (t,y,x) in im_shp coordinates or None.
    Returns (points [B,N,T,2] (x,y), occlusion [B,N,T], expected_dist [B,N,T])."""
    numpy_out = _is_numpy(interp_feature)
    if im_shp is not None and tuple(im_shp[2:4]) != tuple(self.initial_resolution):
      raise ValueError('im_shp must carry initial_resolution')
    qf, g = self._dev(interp_feature), self._dev(feature_grid)
    qp = None if query_points is None else self._dev(query_points)
    B, Q, _ = qf.shape
    _, T, h, w, _ = g.shape
    pts = torch.empty((B, Q, T, 2), device=self.device, dtype=torch.float32)
    occ = torch.empty((B, Q, T), device=self.device, dtype=torch.float32)
    expd = torch.empty((B, Q, T), device=self.device, dtype=torch.float32)
    self._check(self._lib.tapir_tracks_from_cost_volume(
        self._ctx, self._ptr(qf), self._ptr(g), self._ptr(qp), B, Q, T, h, w, self._ptr(pts),
        self._ptr(occ), self._ptr(expd), self._stream()), 'tapir_tracks_from_cost_volume')
    if numpy_out:
      return pts.cpu().numpy(), occ.cpu().numpy(), expd.cpu().numpy()
    return pts, occ, expd

  # ------------------------------------------------------------------ R3 + R4
  def refine_pips(self, target_feature, frame_features, pyramid, pos_guess, occ_guess,
                  expd_guess, orig_hw, last_iter=None, mixer_iter=0.0, resize_hw=None,
                  causal_context=None, get_causal_context=False):
    """tapir_model.py:473-624.  Returns (pos, occ, expd, feats, new_causal_context)."""
    del frame_features, mixer_iter
    assert len(target_feature) == len(pyramid)
    numpy_out = _is_numpy(pos_guess)
    qs = [self._dev(q) for q in target_feature]
    gs = [self._dev(g) for g in pyramid]
    pos, occ, expd = self._dev(pos_guess), self._dev(occ_guess), self._dev(expd_guess)
    last = None if last_iter is None else self._dev(last_iter)
    B, Q, T, _ = pos.shape
    pyr = _ffi.TapirPyramid()
    pyr.n_levels = len(gs)
    for l, (q, g) in enumerate(zip(qs, gs)):
      pyr.query[l] = q.data_ptr(); pyr.grid[l] = g.data_ptr()
      pyr.h[l], pyr.w[l], pyr.C[l] = g.shape[2], g.shape[3], g.shape[4]
    po, oo, eo = torch.empty_like(pos), torch.empty_like(occ), torch.empty_like(expd)
    fo = torch.empty((B, Q, T, HIRES_DIM + LOWRES_DIM), device=self.device, dtype=torch.float32)
    c1i = c2i = c1o = c2o = None
    if causal_context is not None:
      c1i, c2i = self._pack_block_context(causal_context, B * Q)
    if get_causal_context:
      c1o = torch.empty((self.num_mixer_blocks, B * Q, 2, 512), device=self.device)
      c2o = torch.empty((self.num_mixer_blocks, B * Q, 2, 2048), device=self.device)
    self._check(self._lib.tapir_refine_pips(
        self._ctx, ctypes.byref(pyr), B, Q, T, self._ptr(pos), self._ptr(occ), self._ptr(expd),
        self._ptr(last), int(orig_hw[0]), int(orig_hw[1]), int(resize_hw[0]), int(resize_hw[1]),
        self._ptr(po), self._ptr(oo), self._ptr(eo), self._ptr(fo), self._ptr(c1i), self._ptr(c2i),
        self._ptr(c1o), self._ptr(c2o), self._stream()), 'tapir_refine_pips')
    new_cc = {}
    if get_causal_context:
      for i, (n1, n2) in enumerate(causal_block_names(self.num_mixer_blocks, self.haiku_state_names)):
        new_cc[n1] = c1o[i].view(B, Q, 2, 512)
        new_cc[n2] = c2o[i].view(B, Q, 2, 2048)
    outs = (po, oo, eo, fo)
    if numpy_out:
      outs = tuple(o.cpu().numpy() for o in outs)
      new_cc = {k: v.cpu().numpy() for k, v in new_cc.items()}
    return outs + (new_cc,)

  def _pack_block_context(self, ctx: Mapping[str, Any], n: int):
    names = causal_block_names(self.num_mixer_blocks, self.haiku_state_names)
    alt = causal_block_names(self.num_mixer_blocks, not self.haiku_state_names)
    c1, c2 = [], []
    for (n1, n2), (a1, a2) in zip(names, alt):
      v1 = ctx[n1] if n1 in ctx else ctx[a1]
      v2 = ctx[n2] if n2 in ctx else ctx[a2]
      c1.append(self._dev(v1).reshape(n, 2, 512)); c2.append(self._dev(v2).reshape(n, 2, 2048))
    return torch.stack(c1).contiguous(), torch.stack(c2).contiguous()

  # ------------------------------------------------------------------ R1
  def estimate_trajectories(self, video_size: Tuple[int, int], is_training: bool,
                            feature_grids: FeatureGrids, query_features: QueryFeatures,
                            query_points_in_video, query_chunk_size: Optional[int] = None,
                            causal_context=None, get_causal_context: bool = False
                            ) -> Mapping[str, Any]:
    """tapir_model.py:858-1066.  Returns dict(occlusion, tracks, expected_dist
    [, causal_context]) of per-iteration lists, video pixel coordinates."""
    del query_chunk_size
    if causal_context is not None and is_training:
      raise ValueError('Training with causal context is not supported.')
    numpy_out = _is_numpy(feature_grids.lowres[0])
    nl = len(feature_grids.lowres)
    if nl < 2 or nl > _ffi.TAPIR_MAX_LEVELS:
      raise ValueError('feature_grids must hold between 2 and 8 levels')
    lows = [self._dev(x) for x in feature_grids.lowres]
    his = [self._dev(x) for x in feature_grids.hires]
    qls = [self._dev(x) for x in query_features.lowres]
    qhs = [self._dev(x) for x in query_features.hires]
    res = [_res_hw(r) for r in feature_grids.resolutions]
    B, T = lows[0].shape[:2]
    Q = qls[0].shape[1]
    ni = self.num_pips_iter * (nl - 1)
    a = _ffi.TapirTrajArgs()
    a.B, a.Q, a.T, a.n_levels = B, Q, T, nl
    for l in range(nl):
      a.lowres[l] = lows[l].data_ptr(); a.hires[l] = his[l].data_ptr()
      a.lowres_h[l], a.lowres_w[l] = lows[l].shape[2:4]
      a.hires_h[l], a.hires_w[l] = his[l].shape[2:4]
      a.res_h[l], a.res_w[l] = res[l]
      a.q_lowres[l] = qls[l].data_ptr(); a.q_hires[l] = qhs[l].data_ptr()
    qp = None
    if query_points_in_video is not None:
      qp = self._dev(query_points_in_video)
      a.query_points = qp.data_ptr()
    a.video_h, a.video_w = int(video_size[0]), int(video_size[1])
    tracks = torch.empty((ni + 1, B, Q, T, 2), device=self.device, dtype=torch.float32)
    occ = torch.empty((ni + 1, B, Q, T), device=self.device, dtype=torch.float32)
    expd = torch.empty((ni + 1, B, Q, T), device=self.device, dtype=torch.float32)
    a.tracks, a.occlusion, a.expected_dist = tracks.data_ptr(), occ.data_ptr(), expd.data_ptr()
    packed_in = None
    if causal_context is not None:
      packed_in = self._pack_state(causal_context, B * Q, ni)
      a.ctx1_in, a.ctx2_in = packed_in[0].data_ptr(), packed_in[1].data_ptr()
    packed_out = None
    if get_causal_context:
      packed_out = (torch.empty((ni, self.num_mixer_blocks, B * Q, 2, 512), device=self.device),
                    torch.empty((ni, self.num_mixer_blocks, B * Q, 2, 2048), device=self.device))
      a.ctx1_out, a.ctx2_out = packed_out[0].data_ptr(), packed_out[1].data_ptr()
    self._check(self._lib.tapir_estimate_trajectories(self._ctx, ctypes.byref(a), self._stream()),
                'tapir_estimate_trajectories')
    conv = (lambda t: t.cpu().numpy()) if numpy_out else (lambda t: t)
    out = dict(occlusion=[conv(occ[i]) for i in range(ni + 1)],
               tracks=[conv(tracks[i]) for i in range(ni + 1)],
               expected_dist=[conv(expd[i]) for i in range(ni + 1)])
    if get_causal_context:
      out['causal_context'] = self._unpack_state(packed_out, B, Q, numpy_out)
    return out

  def _unpack_state(self, packed, B, Q, numpy_out=False) -> CausalState:
    names = causal_block_names(self.num_mixer_blocks, self.haiku_state_names)
    state = CausalState()
    for i in range(packed[0].shape[0]):
      d = {}
      for j, (n1, n2) in enumerate(names):
        v1, v2 = packed[0][i, j].view(B, Q, 2, 512), packed[1][i, j].view(B, Q, 2, 2048)
        d[n1] = v1.cpu().numpy() if numpy_out else v1
        d[n2] = v2.cpu().numpy() if numpy_out else v2
      state.append(d)
    state.packed = packed
    return state

  def _pack_state(self, causal_context, n: int, ni: int):
    packed = getattr(causal_context, 'packed', None)
    if packed is not None and packed[0].shape[0] == ni and packed[0].shape[2] == n \
        and packed[0].device == self.device:
      return packed
    if len(causal_context) != ni:
      raise ValueError(f'causal_context must have {ni} entries')
    c1, c2 = zip(*[self._pack_block_context(d, n) for d in causal_context])
    return torch.stack(c1).contiguous(), torch.stack(c2).contiguous()

  # ------------------------------------------------------------------ R0
  def __call__(self, video, is_training: bool = False, query_points=None,
               query_chunk_size: Optional[int] = None, get_query_feats: bool = False,
               refinement_resolutions: Optional[List[Tuple[int, int]]] = None,
               feature_grids: Optional[FeatureGrids] = None) -> Mapping[str, Any]:
    """tapir_model.py:1068-1154."""
    if get_query_feats:
      raise ValueError('Get query feats not supported in TAPIR.')
    numpy_out = _is_numpy(video)
    self._staged = []
    if feature_grids is None:
      feature_grids = self.get_feature_grids(video, is_training, refinement_resolutions, _borrow=True)
    elif self.dtype == 'bfloat16' and getattr(feature_grids, 'staged', None):
      self._staged = list(feature_grids.staged)   # (gathered operand copies: tapnet_amd.distributed)
    query_features = self.get_query_features(video, is_training, query_points, feature_grids,
                                             refinement_resolutions)
    fg = FeatureGrids(tuple(self._dev(x) for x in feature_grids.lowres),
                      tuple(self._dev(x) for x in feature_grids.hires), feature_grids.resolutions)
    try:
      # the borrowed grids' operand-type copies (written by the backbone next to them) stand in for the casts of
      # the hot path for exactly this call
      for f32, op, tiled in self._staged:
        self._check(self._lib.tapir_set_staged_grid(self._ctx, f32.data_ptr(), op.data_ptr(),
                                                    tiled.data_ptr() if tiled is not None else None),
                    'tapir_set_staged_grid')
      traj = self.estimate_trajectories(tuple(video.shape[-3:-1]), is_training, fg, query_features,
                                        self._dev(query_points), query_chunk_size)
    finally:
      if self._staged:
        self._lib.tapir_clear_staged_grids(self._ctx)
        self._staged = []
    p = self.num_pips_iter
    conv = (lambda t: t.cpu().numpy()) if numpy_out else (lambda t: t)

    def level_mean(lst):   # mean over the refinement levels (:1142-1152); one level: that iteration as it is
      sel = lst[p::p]
      return sel[0] if len(sel) == 1 else torch.mean(torch.stack(sel), dim=0)

    out = dict(
        occlusion=conv(level_mean(traj['occlusion'])),
        tracks=conv(level_mean(traj['tracks'])),
        expected_dist=conv(level_mean(traj['expected_dist'])),
        unrefined_occlusion=[conv(t) for t in traj['occlusion'][:-1]],
        unrefined_tracks=[conv(t) for t in traj['tracks'][:-1]],
        unrefined_expected_dist=[conv(t) for t in traj['expected_dist'][:-1]],
    )
    return out


  # ------------------------------------------------------------------ R6
  def construct_initial_causal_state(self, num_points: int, num_resolutions: int = 1) -> CausalState:
    """tapir_model.py:1156-1170: zeros [1,N,2,512] / [1,N,2,2048] per block, 4*num_resolutions times."""
    ni = num_resolutions * self.num_pips_iter
    packed = (torch.zeros((ni, self.num_mixer_blocks, num_points, 2, 512), device=self.device),
              torch.zeros((ni, self.num_mixer_blocks, num_points, 2, 2048), device=self.device))
    return self._unpack_state(packed, 1, num_points)

  def update_query_features(self, query_features: QueryFeatures, new_query_features: QueryFeatures,
                            idx_to_update, causal_state=None):
    """tapir_model.py:1172-1203 (functional like the JAX model: inputs are not mutated)."""
    if isinstance(idx_to_update, int):
      idx_to_update = (idx_to_update,)
    idx = torch.as_tensor(np.array(idx_to_update), device=self.device, dtype=torch.long)

    def upd(s1, s2):
      out = self._dev(s1).clone()
      out[:, idx] = self._dev(s2)
      return out

    qf = QueryFeatures(
        lowres=tuple(upd(a, b) for a, b in zip(query_features.lowres, new_query_features.lowres)),
        hires=tuple(upd(a, b) for a, b in zip(query_features.hires, new_query_features.hires)),
        resolutions=query_features.resolutions)
    if causal_state is None:
      return qf
    ni = len(causal_state)
    n = next(iter(causal_state[0].values())).shape[1]
    c1, c2 = self._pack_state(causal_state, n, ni)
    c1, c2 = c1.clone(), c2.clone()
    c1[:, :, idx] = 0.0   # fresh (zero) state for the replaced points
    c2[:, :, idx] = 0.0
    return qf, self._unpack_state((c1, c2), 1, n)


class ParameterizedTAPIR:
  """tapir_model.py:1206-1269: same six entry points as plain callables.  ``params`` is a
  flat torch-named weight dict (or Haiku params, converted by tapnet_amd.weights);
  ``state`` is accepted for signature compatibility and unused (InstanceNorm has none)."""

  def __init__(self, params=None, state=None, tapir_kwargs=None, **engine_kwargs):
    del state
    from tapnet_amd import weights as weights_lib
    kwargs = dict(tapir_kwargs) if tapir_kwargs else {}
    flat = weights_lib.to_torch_names(params) if params is not None else None
    is_haiku = weights_lib.is_haiku_params(params)
    # a Haiku tree means the caller drives the JAX model: its causal-state keys and its antialiased
    # down-resize (tapir_model.py:670; tests/test_jax_reference_pin.py) unless told otherwise
    engine_kwargs.setdefault('jax_antialias_resize', is_haiku)
    self._model = TAPIR(**kwargs, weights=flat, haiku_state_names=is_haiku, **engine_kwargs)
    for fn in ('estimate_trajectories', 'get_query_features', 'get_feature_grids',
               'construct_initial_causal_state', 'update_query_features'):
      setattr(self, fn, getattr(self._model, fn))

  def __call__(self, *args, **kwargs):
    return self._model(*args, **kwargs)
