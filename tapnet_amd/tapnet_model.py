"""Drop-in mirror of ``tapnet.models.tapnet_model.TAPNet`` for the part of TAP-Net that shares the
hot path with TAPIR: the cost-volume head.

  TAPNet.tracks_from_cost_volume(interp_feature_heads, feature_grid_heads, query_points, im_shp)
      tapnet/models/tapnet_model.py:111-171
  TAPNet.__call__(video, is_training, query_points, compute_regression, query_chunk_size,
                  get_query_feats, feature_grid)                                   :173-290

run on the gfx950 engine: the head is the fused cost-volume kernel of TAPIR
(tapnet_amd/csrc/costvol_fused.hpp) with the three TAP-Net differences (no ReLU after the stride-2
convolution, one occlusion logit, softmax temperature 10) -- einsum on the matrix cores into LDS,
convolutions, softmax, soft arg max and the occlusion head in one launch.

Out of scope: the TSM-ResNet backbone (tapnet/models/tsm_resnet.py).  ``__call__``
therefore needs ``feature_grid`` (the L2-normalised [B,T,H/8,W/8,256] grid the reference returns as
``out['feature_grid']``); query features are sampled from it on the GPU (model_utils.interp,
mode='nearest': tapir_get_query_features).  ``num_heads`` 1 (the constructor default; nothing in the reference sets another value), 2 or 4: the head count is
the number of input channels of ``hid1`` in the weights.
Pinned against the reference's own tapnet_model.py executed over numpy stand-ins for jax / haiku
(oracle/make_tapnet_golden.py -> tests/golden/tapnet_head.npz, tests/test_tapnet_reference_pin.py).

Weights: ``{'tapnet_cost_volume_track_mods.<hid1|hid2|hid3|hid4|occ_out>.<weight|bias>': array}`` in
the torch layout of the TAPIR head (hid1 [16,num_heads,3,3], hid2 [1,16,3,3], hid3 [32,16,3,3], hid4 [16,32],
occ_out [1,16]); ``from_haiku_params`` converts the reference's Haiku tree
(``tap_net/~/cost_volume_regression_1`` ...).
"""
from __future__ import annotations

import ctypes
from typing import Any, Mapping, Optional

import numpy as np
import torch

from tapnet_amd import _ffi

HAIKU_NAMES = {   # tapnet_model.py:64-107 (hk.Conv3D kernels are [1,3,3,in,out], hk.Linear [in,out])
    'hid1': 'cost_volume_regression_1', 'hid2': 'cost_volume_regression_2',
    'hid3': 'cost_volume_occlusion_1', 'hid4': 'cost_volume_occlusion_2', 'occ_out': 'occlusion_out',
}


def from_haiku_params(params: Mapping[str, Mapping[str, Any]], scope: str = 'tap_net') -> dict:
  """Haiku params of the reference TAPNet head -> the flat torch-layout dict this module loads.
  The head's modules are constructed in TAPNet.__init__ (tapnet_model.py:79-108), so Haiku files them under
  '<scope>/~/<name>' ('tap_net/~/cost_volume_regression_1'); a tree without the '~' level is accepted too."""
  out = {}
  for short, hk_name in HAIKU_NAMES.items():
    key = f'{scope}/~/{hk_name}'
    mod = params[key] if key in params else params[f'{scope}/{hk_name}']
    w, b = np.asarray(mod['w'], np.float32), np.asarray(mod['b'], np.float32)
    if w.ndim == 5:      # Conv3D [1,3,3,in,out] -> [out,in,3,3]
      w = np.transpose(w[0], (3, 2, 0, 1))
    else:                # Linear [in,out] -> [out,in]
      w = w.T
    out[f'tapnet_cost_volume_track_mods.{short}.weight'] = np.ascontiguousarray(w)
    out[f'tapnet_cost_volume_track_mods.{short}.bias'] = b
  return out


class TAPNet:
  """TAP-Net cost-volume head on the MI355X engine (tapnet_model.py:45-290)."""

  def __init__(self, feature_grid_stride: int = 8, num_heads: int = 1, cross_replica_axis=None,
               num_frames: int = 24, *, weights: Optional[Mapping[str, Any]] = None,
               dtype: str = 'float32', device: Any = None):
    del cross_replica_axis, num_frames
    if num_heads not in (1, 2, 4):
      raise ValueError('the HIP head is built for num_heads 1, 2 or 4')
    if feature_grid_stride != 8:
      raise ValueError('feature_grid_stride must be 8')
    if dtype not in ('float32', 'bfloat16'):
      raise ValueError("dtype must be 'float32' or 'bfloat16'")
    self.feature_grid_stride = feature_grid_stride
    self.num_heads = num_heads
    self.softmax_temperature = 10.0
    if device is None:
      device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type != 'cuda':
      raise RuntimeError('tapnet_amd.TAPNet needs a ROCm GPU (torch device "cuda"); there is no CPU path')
    self.device = torch.device(device)
    if self.device.index is None:
      self.device = torch.device('cuda', torch.cuda.current_device())
    self._lib = _ffi.load_library()
    self._ctx = ctypes.c_void_p()
    cfg = _ffi.TapirCfg(0, 4, 12, 0, 10.0, 256, 256, _ffi.TAPIR_BF16 if dtype == 'bfloat16' else _ffi.TAPIR_F32)
    rc = self._lib.tapir_create(ctypes.byref(self._ctx), ctypes.byref(cfg), self.device.index)
    if rc != _ffi.TAPIR_OK:
      raise ValueError(f'tapir_create failed ({rc})')
    if weights is not None:
      self.load_weights(weights)

  def __del__(self):
    try:
      if getattr(self, '_ctx', None):
        self._lib.tapir_destroy(self._ctx)
        self._ctx = None
    except Exception:
      pass

  def _check(self, rc, what):
    _ffi.check(self._lib, self._ctx, rc, what)

  def _dev(self, x) -> torch.Tensor:
    t = torch.as_tensor(x) if not isinstance(x, torch.Tensor) else x
    return t.to(device=self.device, dtype=torch.float32).contiguous()

  def _stream(self):
    return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def load_weights(self, weights: Mapping[str, Any]):
    for k, v in weights.items():
      if not k.startswith('tapnet_cost_volume_track_mods.'):
        continue
      a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
      a = np.ascontiguousarray(a, dtype=np.float32)
      if k.endswith('hid1.weight') and (a.ndim != 4 or a.shape[1] != self.num_heads):
        raise ValueError(f'hid1.weight {a.shape} does not have num_heads={self.num_heads} input channels')
      shape = (ctypes.c_int64 * a.ndim)(*a.shape)
      self._check(self._lib.tapir_set_weight(self._ctx, k.encode(), a.ctypes.data_as(ctypes.c_void_p),
                                             shape, a.ndim), f'tapir_set_weight({k})')
    self._check(self._lib.tapir_finalize_weights(self._ctx), 'tapir_finalize_weights')

  def tracks_from_cost_volume(self, interp_feature_heads, feature_grid_heads, query_points, im_shp=None):
    """tapnet_model.py:111-171.  interp_feature_heads [B,N,C,1], feature_grid_heads [B,T,h,w,C,1]
    (or without the trailing heads axis), query_points [B,N,3] (t,y,x) in im_shp pixels or None,
    im_shp [B,T,H,W,3].  Returns (points [B,N,T,2] (x,y) in im_shp pixels, occlusion [B,N,T])."""
    numpy_out = isinstance(interp_feature_heads, np.ndarray)
    qf, g = self._dev(interp_feature_heads), self._dev(feature_grid_heads)
    if qf.ndim == 4:     # [B,N,C/d,d] and [B,T,h,w,C/d,d]: 'b n (c d)' is the same memory as [B,N,C]
      if qf.shape[-1] != self.num_heads:
        raise ValueError(f'expected {self.num_heads} heads, got {qf.shape[-1]}')
      qf, g = qf.reshape(*qf.shape[:2], -1), g.reshape(*g.shape[:4], -1)
    B, Q, C = qf.shape
    _, T, h, w, _ = g.shape
    if C != 256:
      raise ValueError('feature dimension must be 256 (tsm_resnet_unit_2 of TSM-ResNet-18)')
    H, W = (h * self.feature_grid_stride, w * self.feature_grid_stride) if im_shp is None else (int(im_shp[2]), int(im_shp[3]))
    qp = None
    if query_points is not None:   # the kernel works in its 256x256 'initial resolution' frame
      qp = self._dev(query_points) * torch.tensor([1.0, 256.0 / H, 256.0 / W], device=self.device)
      qp = qp.contiguous()
    pts = torch.empty((B, Q, T, 2), device=self.device, dtype=torch.float32)
    occ = torch.empty((B, Q, T), device=self.device, dtype=torch.float32)
    self._check(self._lib.tapir_tapnet_tracks_from_cost_volume(
        self._ctx, qf.data_ptr(), g.data_ptr(), None if qp is None else qp.data_ptr(), B, Q, T, h, w,
        pts.data_ptr(), occ.data_ptr(), self._stream()), 'tapir_tapnet_tracks_from_cost_volume')
    pts = pts * torch.tensor([W / 256.0, H / 256.0], device=self.device)
    if numpy_out:
      return pts.cpu().numpy(), occ.cpu().numpy()
    return pts, occ

  def cycle_consistency_tracks(self, query_feats, feature_grid, query_points, im_shp,
                               softmax_temperature: Optional[float] = None, dist_threshold: float = 48.0,
                               return_inverse_tracks: bool = False):
    """The forward-backward cycle-consistency tracker of TAP-Net's evaluation path
    (tapnet/training/supervised_point_prediction.py:443-546; prediction_algo other than 'cost_volume_regressor'): no
    learned head.  query_feats [B,N,C] and feature_grid [B,T,h,w,C] as returned by __call__(get_query_feats=True)
    (out['query_feats'], out['feature_grid']), query_points [B,N,3] (t,y,x) in im_shp pixels, im_shp [B,T,H,W,3].
    Returns (tracks [B,N,T,2] (x,y) px, occlusion logits [B,N,T] = +10 / -10)."""
    numpy_out = isinstance(query_feats, np.ndarray)
    qf, g, qp = self._dev(query_feats), self._dev(feature_grid), self._dev(query_points)
    B, Q, C = qf.shape
    _, T, h, w, _ = g.shape
    if C != 256:
      raise ValueError('feature dimension must be 256 (tsm_resnet_unit_2 of TSM-ResNet-18)')
    H, W = int(im_shp[2]), int(im_shp[3])
    temp = float(self.softmax_temperature if softmax_temperature is None else softmax_temperature)
    pts = torch.empty((B, Q, T, 2), device=self.device, dtype=torch.float32)
    occ = torch.empty((B, Q, T), device=self.device, dtype=torch.float32)
    inv = torch.empty((B, Q, T, 2), device=self.device, dtype=torch.float32)
    self._check(self._lib.tapir_cycle_consistency_tracks(
        self._ctx, qf.data_ptr(), g.data_ptr(), qp.data_ptr(), B, Q, T, h, w, H, W, temp, float(dist_threshold),
        pts.data_ptr(), occ.data_ptr(), inv.data_ptr(), self._stream()), 'tapir_cycle_consistency_tracks')
    outs = (pts, occ, inv) if return_inverse_tracks else (pts, occ)
    return tuple(t.cpu().numpy() for t in outs) if numpy_out else outs

  def __call__(self, video, is_training: bool = False, query_points=None, compute_regression: bool = True,
               query_chunk_size: Optional[int] = None, get_query_feats: bool = False, feature_grid=None):
    """tapnet_model.py:173-290 from a precomputed feature grid.  ``video`` is used for its shape only
    (it may be a shape tuple); ``query_chunk_size`` is accepted and ignored (nothing is chunked: the
    volume never exists)."""
    del is_training, query_chunk_size
    if feature_grid is None:
      raise NotImplementedError('the TSM-ResNet backbone is out of scope: pass feature_grid '
                                '(the L2-normalised grid the reference returns as out["feature_grid"])')
    numpy_out = isinstance(feature_grid, np.ndarray)
    shape = tuple(video.shape) if hasattr(video, 'shape') else tuple(video)
    fg = self._dev(feature_grid)
    qp = self._dev(query_points)
    B, Q = qp.shape[:2]
    _, T, h, w, C = fg.shape
    interp = torch.empty((B, Q, C), device=self.device, dtype=torch.float32)
    self._check(self._lib.tapir_get_query_features(self._ctx, fg.data_ptr(), qp.data_ptr(), B, Q, T, h, w, C,
                                                   int(shape[2]), int(shape[3]), interp.data_ptr(),
                                                   self._stream()), 'tapir_get_query_features')
    conv = (lambda t: t.cpu().numpy()) if numpy_out else (lambda t: t)
    out = {'feature_grid': feature_grid}
    if get_query_feats:
      out['query_feats'] = conv(interp)
    if compute_regression:
      d = self.num_heads
      pts, occ = self.tracks_from_cost_volume(interp.reshape(B, Q, C // d, d), fg.reshape(B, T, h, w, C // d, d), qp,
                                              im_shp=shape)
      out['occlusion'] = conv(occ)
      out['tracks'] = conv(pts)
    return out
