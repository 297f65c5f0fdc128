"""Feature backbone (R7): HIP kernels + the remaining convolutions on PyTorch-ROCm / MIOpen
(north_star keeps the ResNet convolutions on PyTorch; SURVEY.md 8 f1 moves them to HIP).

Re-statement of TAPIR.get_feature_grids (tapnet/models/tapir_model.py:626-729):
bilinear resize -> ResNet-v2 with InstanceNorm (tapnet/models/resnet.py:150-257,
305-471; strides (1,2,2,1), channels (64,128,256,256), 2 blocks per group,
projection shortcut on the first block of each group, 7x7/2 stem) -> optional
ExtraConvs (tapir_model.py:159-186) on the low-res map -> per-pixel L2
normalisation (:709-720).  Convolutions use XLA "SAME" padding (extra pad on the
high side for stride 2).  Weights are looked up by the reference's torch
state_dict names (tapnet/torch/nets.py).

Runs channels-last (NHWC) so that the feature grids leave in the
[B,T,h,w,C] layout the HIP kernels read, with no transpose.

GPU only.  The convolutions of the ResNet blocks (3x3 and 1x1, stride 1 and 2) are the fused HIP kernel
of tapnet_amd/csrc/conv_fused.hpp -- InstanceNorm + ReLU in its operand load, residual add and the next
norm's statistics in its epilogue (tapir_conv_fused; bf16 MFMA in bf16 contexts, exact-f32 MFMA in f32
contexts) and so is the 7x7 stem (tapir_stem_conv); only the ExtraConvs of BootsTAPIR run in PyTorch
(MIOpen / CK implicit-GEMM, NHWC); what is left between the kernels (the merge of the tile summaries,
the final L2 normalisation; the statistics / normalise kernels behind a convolution whose shape does
not fit the HIP kernel and goes to MIOpen) are the HIP kernels of csrc/backbone.hpp
(tapir_inorm_stats / tapir_inorm_relu / tapir_l2_normalize).  From the third call with one shape on, a clip's launches are replayed
from a hipGraph.  There is no CPU path here; the plain PyTorch restatement used by the CPU tests and
by bench.py's cpu_baseline lives in oracle/backbone_torch.py.
"""
from __future__ import annotations

import os
from typing import Dict, NamedTuple, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


class _Stats(NamedTuple):
  """InstanceNorm summaries of a tensor: part [N, slabs, C, 2] (mean, M2) per slab of per_s pixels
  (0: ceil(HW / slabs))."""
  part: torch.Tensor
  slabs: int
  per_s: int
  ss: Optional[torch.Tensor] = None      # the (a, b) pairs of norm `ss_norm`, merged by the producing launch itself
  ss_norm: Optional[str] = None


def _same_pad(x: torch.Tensor, k: int, stride: int) -> torch.Tensor:
  """XLA SAME padding for an NCHW tensor: total = max((ceil(n/s)-1)*s+k-n, 0), low = total//2."""
  h, w = x.shape[-2:]
  th = max((-(-h // stride) - 1) * stride + k - h, 0)
  tw = max((-(-w // stride) - 1) * stride + k - w, 0)
  if th == 0 and tw == 0:
    return x
  return F.pad(x, (tw // 2, tw - tw // 2, th // 2, th - th // 2))


class Backbone:
  """Holds the backbone weights on `device` in `dtype` (float32 or bfloat16)."""

  def __init__(self, weights: Dict[str, torch.Tensor], extra_convs: bool, device,
               dtype: torch.dtype = torch.float32,
               blocks_per_group: Sequence[int] = (2, 2, 2, 2), engine=None):
    """engine: (ctypes library, context) of libtapir_hip.so -- required."""
    self.device = torch.device(device)
    self.engine = engine
    self.last_staged = None
    if self.device.type != 'cuda' or engine is None:
      raise RuntimeError('tapnet_amd.backbone.Backbone needs a ROCm GPU and the HIP engine '
                         '(libtapir_hip.so); there is no CPU path')
    # MIOpen exhaustive solver search per convolution shape (first call only): the default
    # heuristic picks atomic split-K implicit-GEMM kernels that need a zero-fill pass per call;
    # measured 3.12 -> 2.66 ms per 48-frame clip.  The flag is process-wide in PyTorch, so it is set
    # only around this backbone's own convolutions (features()) and restored afterwards.
    self.miopen_exhaustive_search = True
    # Frames are independent through the whole backbone, so a clip can be cut into `streams` groups of
    # frames that run on separate HIP streams: the memory-bound phases of one group's kernels (tile
    # staging, epilogues, the glue kernels) then overlap the MFMA-bound phases of the other's instead of
    # alternating with them (1.86 -> 1.64 / 1.57 ms with 2 / 4 streams for a 48-frame clip, bit-identical: every kernel here is
    # independent of how many frames a launch covers).  1 = everything on the caller's stream; applies
    # to clips of at least 8 frames per stream.
    # With BootsTAPIR's ExtraConvs (large, MFMA-bound launches that fill the chip on their own) two groups are 3.5 %
    # faster than four (3.49-3.54 against 3.64-3.69 ms per clip, tools/exp_streams.py, same box).
    self.streams = (2 if extra_convs else 4) if dtype == torch.bfloat16 else 1
    if os.environ.get('TAPIR_BACKBONE_STREAMS'):   # (A/B measurements, tools/bench_backbone.py)
      self.streams = int(os.environ['TAPIR_BACKBONE_STREAMS'])
    self._side_streams = []
    self._lane = 0
    # 'auto': the convolutions of the ResNet blocks (3x3 and 1x1, stride 1 and 2: everything but the 7x7
    # stem) run as the fused HIP kernel of csrc/conv_fused.hpp where the shape fits it (bf16
    # contexts); 'miopen': every convolution through MIOpen (the A/B switch, and the f32 path).
    self.conv_mode = 'auto'     # 'auto' | 'hip' (any number of frames) | 'miopen'
    self.hip_min_frames = 4 if dtype == torch.bfloat16 else 1   # (f32 = the parity build: nothing is timed)
    if os.environ.get('TAPIR_HIP_MIN_FRAMES'):   # (A/B measurements of the online step: 1 = the HIP convolutions for a single frame too)
      self.hip_min_frames = int(os.environ['TAPIR_HIP_MIN_FRAMES'])
    self._hip_now = False
    # clips of fewer than `small_max_frames` frames (the online model's single frame): the K-split form of the block
    # convolutions (csrc/conv_small.hpp) instead of the library's kernels + glue; decided per features() call from the
    # WHOLE clip like _hip_now (the two HIP forms differ in summation order).  TAPIR_CONV_SMALL=0: off (A/B)
    self.small_convs = dtype == torch.bfloat16 and os.environ.get('TAPIR_CONV_SMALL', '1') != '0'
    self.small_max_frames = 4
    self._small_now = False
    self._clip_frames = 0       # frames of the whole clip of the current features() call (kernel choices follow it, never a shard)
    # which kinds of block convolution take the HIP kernel in 'auto' mode (the others stay on MIOpen)
    self.hip_convs = {'stem', 'conv_0', 'conv_1', 'conv_0_s2', 'proj_conv', 'proj_conv_s2'}
    # clips of at least this many frames replay their launches from a hipGraph from the third call with
    # the same shape on (features()); 0 = always launch eagerly (the f32 build: parity tests, nothing timed)
    self.graph_min_frames = 8 if dtype == torch.bfloat16 else 0
    self._graphs: Dict[tuple, dict] = {}
    self._plans: Dict[tuple, Optional[tuple]] = {}
    self._stem_ws = None
    self._wstream: Dict[str, int] = {}
    self._xstream: Dict[tuple, int] = {}     # ExtraConvs weight packs, by (conv name, input channels per chunk)
    self._xhost: Dict[str, 'np.ndarray'] = {}
    # the producing convolution merges the next InstanceNorm's (a, b) pairs itself (tapir_conv_fused_nn) instead of a
    # merge launch in front of the consuming convolution; '0' = the separate launches (the A/B switch)
    self.fuse_finalize = os.environ.get('TAPIR_FUSE_FINALIZE', '1') != '0'
    # conv_0 and proj_conv of a group's first block in one launch (they read the same normalised tensor: one staging of
    # it instead of two, 4 launches off a frame group's chain); '0' = two launches (the A/B switch)
    self.fuse_proj = os.environ.get('TAPIR_FUSE_PROJ', '1') != '0'
    self._wdual = {}
    self.extra_convs_mode = os.environ.get('TAPIR_EXTRA_CONVS', 'hip')   # 'hip' | 'torch' (MIOpen convolutions + torch glue: the A/B switch)
    self._bufs: Dict[tuple, torch.Tensor] = {}
    self.dtype = dtype
    self.extra_convs = extra_convs
    self.blocks_per_group = tuple(blocks_per_group)
    self.w: Dict[str, torch.Tensor] = {}
    for k, v in weights.items():
      if not (k.startswith('resnet_torch.') or k.startswith('extra_convs.')):
        continue
      t = torch.as_tensor(v).to(self.device)
      if t.ndim == 4:   # conv kernels: compute dtype, channels-last
        t = t.to(dtype).contiguous(memory_format=torch.channels_last)
      else:             # norm scales / biases stay f32
        t = t.float()
      self.w[k] = t
    if dtype in (torch.bfloat16, torch.float32):
      import ctypes
      import numpy as np
      lib, ctx = engine
      for k, v in weights.items():
        if not (k.startswith('resnet_torch.block_groups.') and
                k.endswith(('conv_0.weight', 'conv_1.weight', 'proj_conv.weight'))):
          continue
        a = np.ascontiguousarray(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
        if a.ndim != 4 or a.shape[2] != a.shape[3]:
          continue
        h = ctypes.c_void_p()
        rc = lib.tapir_conv_pack(ctx, a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1], a.shape[2],
                                 ctypes.byref(h))
        if rc == 0:       # (TAPIR_ERR_UNSUPPORTED: that convolution stays on MIOpen)
          self._wstream[k[:-len('.weight')]] = (h.value, a.shape[1], a.shape[0], a.shape[2])
      # conv_0 + proj_conv of a group's first block as ONE launch (csrc/conv_fused.hpp DUAL): a combined weight stream
      self._wdual: Dict[str, tuple] = {}
      strides = (1, 2, 2, 1)
      for g in range(4):
        p = f'resnet_torch.block_groups.{g}.blocks.0.'
        w3, w1 = weights.get(p + 'conv_0.weight'), weights.get(p + 'proj_conv.weight')
        if w3 is None or w1 is None or (p + 'conv_0') not in self._wstream or (p + 'proj_conv') not in self._wstream:
          continue
        a3 = np.ascontiguousarray(w3.detach().cpu().numpy() if isinstance(w3, torch.Tensor) else w3, dtype=np.float32)
        a1 = np.ascontiguousarray(w1.detach().cpu().numpy() if isinstance(w1, torch.Tensor) else w1, dtype=np.float32)
        if a3.shape[2:] != (3, 3) or a1.shape[2:] != (1, 1) or a3.shape[:2] != a1.shape[:2]:
          continue
        h = ctypes.c_void_p()
        rc = lib.tapir_conv_pack_dual(ctx, a3.ctypes.data_as(ctypes.c_void_p), a1.ctypes.data_as(ctypes.c_void_p),
                                      a3.shape[0], a3.shape[1], strides[g], ctypes.byref(h))
        if rc == 0:
          self._wdual[p] = (h.value, a3.shape[1], a3.shape[0], strides[g])
      for k, v in weights.items():          # ExtraConvs kernels: packed on first use (the chunking depends on the map)
        if k.startswith('extra_convs.') and k.endswith(('conv.weight', 'conv_1.weight')):
          self._xhost[k[:-len('.weight')]] = np.ascontiguousarray(
              v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v, dtype=np.float32)
      k = 'resnet_torch.initial_conv.weight'
      a = weights.get(k)
      if a is not None:
        a = np.ascontiguousarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a, dtype=np.float32)
        if a.shape == (64, 3, 7, 7):
          h = ctypes.c_void_p()
          if lib.tapir_stem_pack(ctx, a.ctypes.data_as(ctypes.c_void_p), ctypes.byref(h)) == 0:
            self._stem_ws = h.value
    need = ['resnet_torch.initial_conv.weight']
    if extra_convs:
      need.append('extra_convs.blocks.0.conv.weight')
    for n in need:
      if n not in self.w:
        raise KeyError(f'backbone weight missing: {n}')

  def describe(self, frames: int) -> str:
    """What runs for a clip of `frames` frames (bench.py prints it in its JSON line)."""
    hip = self._use_hip_convs(frames)
    missing = sorted(k for k in ('stem', 'conv_0', 'conv_1', 'conv_0_s2', 'proj_conv', 'proj_conv_s2')
                     if k not in self.hip_convs)
    if hip and self._stem_ws is not None and self._wstream and not missing:
      s = ('HIP: 7x7 stem + every ResNet block convolution (3x3 / 1x1, stride 1 / 2) as fused implicit-GEMM MFMA '
           'kernels (InstanceNorm+ReLU in the operand load, residual add + next-norm statistics in the epilogue' +
           (', the next norm\'s (a, b) pairs merged by the last-arriving workgroup of each image), HIP L2-normalise kernel'
            if self.fuse_finalize else '), HIP finalize / L2-normalise kernels') +
           ('; conv_0 + proj_conv of a group\'s first block in one launch'
            if (self.fuse_proj and self._wdual and not self._small_ok(frames)) else '') +
           ('; few-frame form of the block convolutions: a workgroup per (row tile, 64 output channels), the waves split the taps'
            if self._small_ok(frames) else ''))
    elif hip:
      s = 'HIP fused convolutions except ' + ', '.join(missing or ['(unpacked shapes)']) + ' (MIOpen) + HIP norm kernels'
    else:
      s = 'MIOpen convolutions + HIP InstanceNorm / add / L2 kernels'
    if self.extra_convs:
      s += '; ExtraConvs: ' + self._extra_convs_impl(hip)
    if self.graph_min_frames and frames >= self.graph_min_frames:
      s += f'; {max(1, min(int(self.streams), frames // 8))} streams, hipGraph replay'
    return s

  def _extra_convs_impl(self, hip=True) -> str:
    if hip and self.extra_convs_mode == 'hip' and self._xhost:
      return ('HIP: LayerNorm kernel + 3x3 implicit-GEMM MFMA kernels (256 -> 1024 with bias + GELU, 1024 -> 256 '
              'with bias + skip in the epilogue)')
    return 'PyTorch-ROCm / MIOpen convolutions + torch LayerNorm / GELU'

  def close(self):
    """Releases the packed weight streams this backbone owns in the engine context (tapir_conv_free)
    and drops its captured graphs, which hold their addresses."""
    self._graphs = {}
    eng = getattr(self, 'engine', None)
    if eng is None:
      return
    lib, ctx = eng
    for h in ([v[0] for v in getattr(self, '_wstream', {}).values()] + [getattr(self, '_stem_ws', None)] +
              list(getattr(self, '_xstream', {}).values()) + [v[0] for v in getattr(self, '_wdual', {}).values()]):
      if h:
        lib.tapir_conv_free(ctx, h)
    self._wstream, self._stem_ws, self._xstream, self._wdual = {}, None, {}, {}

  # -- ExtraConvs (BootsTAPIR): small 32x32 maps, PyTorch ops on the GPU -----
  def _conv(self, x, name, stride=1, bias=False):
    w = self.w[name + '.weight']
    b = self.w[name + '.bias'].to(x.dtype) if bias else None
    return F.conv2d(_same_pad(x, w.shape[-1], stride), w, b, stride=stride)

  def _extra_convs(self, x):
    # x NCHW channels-last; LayerNorm over channels with scale+offset (tapir_model.py:176)
    for n in range(5):
      p = f'extra_convs.blocks.{n}.'
      xl = x.permute(0, 2, 3, 1)
      xl = F.layer_norm(xl.float(), (xl.shape[-1],), self.w[p + 'layer_norm.weight'],
                        self.w[p + 'layer_norm.bias'], eps=1e-5).to(x.dtype)
      x = xl.permute(0, 3, 1, 2)
      r = F.gelu(self._conv(x, p + 'conv', 1, bias=True), approximate='tanh')
      x = x + self._conv(r, p + 'conv_1', 1, bias=True)
    return x

  # -- ExtraConvs as HIP kernels (csrc/extra_convs.hpp) ----------------------
  def _xplan(self, h, w, cin, cout):
    """(input channels per LDS chunk, kernel form) for an ExtraConvs convolution on an [h, w] map, or None.  The form
    (64 / 128 pixels per workgroup) follows the frame count of the WHOLE clip (csrc/extra_convs.hpp xconv_plan): the two
    forms add the input channels in different orders, so a chunk or a rank's shard must run what the whole clip runs."""
    frames = int(self._clip_frames)
    key = ('x', h, w, cin, cout, frames)
    if key not in self._plans:
      import ctypes
      lib, ctx = self.engine
      rows, tiles, cch, form = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
      ok = lib.tapir_xconv_plan_frames(ctx, frames, h, w, cin, cout, ctypes.byref(rows), ctypes.byref(tiles),
                                       ctypes.byref(cch), ctypes.byref(form)) == 0
      self._plans[key] = (cch.value, form.value) if ok else None
    return self._plans[key]

  def _xpack(self, name, cch):
    key = (name, cch)
    if key not in self._xstream:
      import ctypes
      lib, ctx = self.engine
      a = self._xhost[name]
      h = ctypes.c_void_p()
      self._check(lib.tapir_xconv_pack(ctx, a.ctypes.data_as(ctypes.c_void_p), a.shape[0], a.shape[1], cch,
                                       ctypes.byref(h)), 'tapir_xconv_pack')
      self._xstream[key] = h.value
    return self._xstream[key]

  def _extra_convs_hip(self, x):
    """tapir_model.py:159-186 on an NHWC map [n,h,w,256] of the element type: per block one LayerNorm kernel
    and two convolution kernels (bias + GELU / bias + skip in their epilogues).  None: shape not covered."""
    lib, ctx = self.engine
    n, h, w, c = x.shape
    c1, c2 = self._xplan(h, w, c, 4 * c), self._xplan(h, w, 4 * c, c)
    if c1 is None or c2 is None or not self._xhost:
      return None
    st = self._stream()
    for blk in range(5):
      p = f'extra_convs.blocks.{blk}.'
      y = self._buf(('xln', n, h, w, c), (n, h, w, c), self.dtype)
      self._check(lib.tapir_layernorm_affine(ctx, x.data_ptr(), self.w[p + 'layer_norm.weight'].data_ptr(),
                                             self.w[p + 'layer_norm.bias'].data_ptr(), y.data_ptr(), n * h * w, c, st),
                  'tapir_layernorm_affine')
      r = self._buf(('xhid', n, h, w, 4 * c), (n, h, w, 4 * c), self.dtype)
      self._check(lib.tapir_xconv_nt(ctx, y.data_ptr(), self._xpack(p + 'conv', c1[0]), self.w[p + 'conv.bias'].data_ptr(),
                                     None, r.data_ptr(), n, h, w, c, 4 * c, 1, c1[1], st), 'tapir_xconv_nt')
      out = self._buf(('xout', blk & 1, n, h, w, c), (n, h, w, c), self.dtype)
      self._check(lib.tapir_xconv_nt(ctx, r.data_ptr(), self._xpack(p + 'conv_1', c2[0]), self.w[p + 'conv_1.bias'].data_ptr(),
                                     y.data_ptr(), out.data_ptr(), n, h, w, 4 * c, c, 0, c2[1], st), 'tapir_xconv_nt')
      x = out
    return x

  # -- GPU path: MIOpen convolutions + HIP glue kernels ---------------------
  def _buf(self, key, shape, dtype, zero=False):
    """Stream-ordered scratch reused across layers and calls (a zero border stays zero: the
    kernels never write it)."""
    key = (self._lane,) + tuple(key)    # groups of frames in flight on different streams do not share scratch
    t = self._bufs.get(key)
    if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
      t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
      self._bufs[key] = t
    return t

  def _check(self, rc, what):
    from tapnet_amd import _ffi
    if rc != 0:
      # a launch sequence that stops half way may leave arrival counters of the in-launch InstanceNorm merge
      # (csrc/conv_fused.hpp) non-zero, and every later merge of that buffer would then fire early: start clean
      for k, t in self._bufs.items():
        if len(k) > 1 and k[1] == 'arrive':
          t.zero_()
    _ffi.check(self.engine[0], self.engine[1], rc, what)

  def _stream(self):
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def _hip_stats(self, a: torch.Tensor, b: Optional[torch.Tensor] = None) -> '_Stats':
    """InstanceNorm summaries of a (NHWC), or of a + b with the sum written over a."""
    lib, ctx = self.engine
    n, h, w, c = a.shape
    # (independent of the number of frames: the summation order, and with it the result, must not
    # depend on how a clip is cut into groups of frames or sharded over ranks)
    slabs = max(1, min(h * w // 64, 32))   # the finalize kernel walks them serially
    part = self._buf(('part', n, slabs, c), (n, slabs, c, 2), torch.float32)
    self._check(lib.tapir_inorm_stats(ctx, a.data_ptr(), b.data_ptr() if b is not None else None,
                                      a.data_ptr() if b is not None else None, part.data_ptr(),
                                      n, h * w, c, slabs, self._stream()), 'tapir_inorm_stats')
    return _Stats(part, slabs, 0)

  def _hip_norm_relu(self, x, st: '_Stats', name, tag, pad=False, sub=False):
    lib, ctx = self.engine
    n, h, w, c = x.shape
    oh, ow = (h + 1, w + 1) if pad else (h, w)
    y = self._buf(('y', tag, n, oh, ow, c), (n, oh, ow, c), self.dtype, zero=pad)
    ys = self._buf(('ysub', tag, n, h // 2, w // 2, c), (n, h // 2, w // 2, c), self.dtype) if sub else None
    ss = self._buf(('ss', n, c), (n, c, 2), torch.float32)
    self._check(lib.tapir_inorm_relu(ctx, x.data_ptr(), st.part.data_ptr(),
                                     self.w[name + '.weight'].data_ptr(), self.w[name + '.bias'].data_ptr(),
                                     ss.data_ptr(), y.data_ptr(), ys.data_ptr() if sub else None, n, h, w, c, st.slabs,
                                     st.per_s, oh, ow, self._stream()), 'tapir_inorm_relu')
    return y, ys

  def _hip_conv(self, x_nhwc, name, stride=1, padding=0):
    """NHWC tensor -> NHWC tensor through MIOpen (channels-last view, no copies)."""
    y = F.conv2d(x_nhwc.permute(0, 3, 1, 2), self.w[name + '.weight'], None, stride=stride,
                 padding=padding)
    return y.permute(0, 2, 3, 1).contiguous()   # no-op for a channels-last result

  # -- the block convolutions: one HIP kernel each (csrc/conv_fused.hpp) ------------------------------
  def _plan(self, h, w, cin, cout, ks, stride):
    """(rows per tile, tiles per image) of the fused convolution, or None: that shape stays on MIOpen."""
    key = (self._small_now, h, w, cin, cout, ks, stride)   # (the few-frame form has its own tile geometry)
    if key not in self._plans:
      import ctypes
      lib, ctx = self.engine
      rows, tiles = ctypes.c_int(), ctypes.c_int()
      ok = lib.tapir_conv_plan(ctx, h, w, cin, cout, ks, stride, ctypes.byref(rows), ctypes.byref(tiles)) == 0
      self._plans[key] = (rows.value, tiles.value) if ok else None
    return self._plans[key]

  def _use_hip_convs(self, n):
    """'auto': the HIP convolutions from `hip_min_frames` frames per CALL on -- a single frame (the
    online model) gives them too few workgroups per launch, the library's kernels win there.  Decided
    once per features() call from the whole clip (`global_frames` when the clip is sharded over ranks,
    tapnet_amd.distributed), never per group of frames: a short last chunk or a small frame shard must
    run the same kernels as the rest of the clip, or sharded and unsharded results differ by rounding."""
    return self.conv_mode == 'hip' or (self.conv_mode == 'auto' and (n >= self.hip_min_frames or self._small_ok(n)))

  def _small_ok(self, n):
    return bool(self.small_convs and self.engine is not None and n < self.small_max_frames and self.conv_mode != 'miopen')

  def _fusable(self, conv_name, h, w, stride):
    kind = conv_name.rsplit('.', 1)[-1] + ('_s2' if stride == 2 else '')
    if not self._hip_now or conv_name not in self._wstream or kind not in self.hip_convs:
      return False
    _, cin, cout, ks = self._wstream[conv_name]
    return self._plan(h, w, cin, cout, ks, stride) is not None

  def _next_norm(self, next_norm, n, cout, tag):
    """tapir_next_norm for the launch that produces the input of `next_norm` (None: no in-launch merge)."""
    # (few-frame form: the CONSUMER merges the summaries in its prologue, csrc/conv_small.hpp -- nothing at the producer's tail)
    if not (self.fuse_finalize and next_norm is not None) or self._small_now:
      return None, None
    from tapnet_amd import _ffi
    ssn = self._buf(('ssn', tag, n, cout), (n, cout, 2), torch.float32)
    arrive = self._buf(('arrive', n), (n,), torch.int32, zero=True)     # zero once; every launch leaves it zero
    nn = _ffi.TapirNextNorm(self.w[next_norm + '.weight'].data_ptr(), self.w[next_norm + '.bias'].data_ptr(),
                            ssn.data_ptr(), arrive.data_ptr())
    return nn, ssn

  def _fused_conv_dual(self, x, st: '_Stats', p, tag, stride, next_norm):
    """conv_0 and proj_conv of block `p` (resnet.py:232-247: both read relu(bn_0(x))) in ONE launch
    (tapir_conv_fused_dual_nn) -> (conv_0's raw output, the projected shortcut, conv_0's summaries)."""
    lib, ctx = self.engine
    n, h, w, _ = x.shape
    ws, cin, cout, _ = self._wdual[p]
    rows, tiles = self._plan(h, w, cin, cout, 3, stride)
    ho, wo = -(-h // stride), -(-w // stride)
    y = self._buf(('fy', tag + 'c', n, ho, wo, cout), (n, ho, wo, cout), self.dtype)
    yp = self._buf(('fy', tag + 'p', n, ho, wo, cout), (n, ho, wo, cout), self.dtype)
    part = self._buf(('fpart', tag + 'c', n, tiles, cout), (n, tiles, cout, 2), torch.float32)
    norm_name = p + 'bn_0'
    merged = st.ss is not None and st.ss_norm == norm_name     # the producer of x merged this norm's pairs already
    ss = st.ss if merged else self._buf(('ss', n, cin), (n, cin, 2), torch.float32)
    assert y.data_ptr() != x.data_ptr() and yp.data_ptr() != x.data_ptr()
    import ctypes
    nn, ssn = self._next_norm(next_norm, n, cout, tag + 'c')
    self._check(lib.tapir_conv_fused_dual_nn(
        ctx, x.data_ptr(), None if merged else st.part.data_ptr(), st.slabs, st.per_s,
        self.w[norm_name + '.weight'].data_ptr(), self.w[norm_name + '.bias'].data_ptr(), ss.data_ptr(), ws,
        y.data_ptr(), yp.data_ptr(), part.data_ptr(), n, h, w, cin, cout, stride,
        ctypes.byref(nn) if nn is not None else None, self._stream()), 'tapir_conv_fused_dual_nn')
    return y, yp, _Stats(part, tiles, rows * wo, ssn, next_norm if nn is not None else None)

  def _fused_conv(self, x, st: '_Stats', norm_name, conv_name, shortcut, tag, stride=1, stats=True, reuse_ss=False,
                  next_norm=None):
    """conv(relu(instance_norm(x))) (+ shortcut) and the summaries of the result, one launch
    (+ the tiny merge of the input summaries; reuse_ss: the previous call on this stream merged the same
    summaries with the same norm -- conv_0 after proj_conv -- and its (a, b) pairs are still in the scratch)."""
    lib, ctx = self.engine
    n, h, w, _ = x.shape
    ws, cin, cout, ks = self._wstream[conv_name]
    rows, tiles = self._plan(h, w, cin, cout, ks, stride)
    ho, wo = -(-h // stride), -(-w // stride)
    y = self._buf(('fy', tag, n, ho, wo, cout), (n, ho, wo, cout), self.dtype)
    part = self._buf(('fpart', tag, n, tiles, cout), (n, tiles, cout, 2), torch.float32) if stats else None
    merged = st.ss is not None and st.ss_norm == norm_name     # the producer of x merged this norm's pairs already
    ss = st.ss if merged else self._buf(('ss', n, cin), (n, cin, 2), torch.float32)
    assert y.data_ptr() != x.data_ptr() and (shortcut is None or y.data_ptr() != shortcut.data_ptr())
    import ctypes
    nn, ssn = self._next_norm(next_norm if stats else None, n, cout, tag)
    self._check(lib.tapir_conv_fused_nn(
        ctx, x.data_ptr(), None if (reuse_ss or merged) else st.part.data_ptr(), st.slabs, st.per_s,
        self.w[norm_name + '.weight'].data_ptr(), self.w[norm_name + '.bias'].data_ptr(), ss.data_ptr(), ws,
        shortcut.data_ptr() if shortcut is not None else None, y.data_ptr(),
        part.data_ptr() if stats else None, n, h, w, cin, cout, ks, stride,
        ctypes.byref(nn) if nn is not None else None, self._stream()), 'tapir_conv_fused_nn')
    return y, (_Stats(part, tiles, rows * wo, ssn, next_norm if nn is not None else None) if stats else None)

  def _hip_block(self, x, st: '_Stats', p, stride, use_projection, tag, parity, next_norm=None):
    """One BlockV2 (resnet.py:185-257).  x: the raw residual stream, st: its InstanceNorm summaries.
    Every convolution that has a HIP kernel for its shape reads x (or conv_0's raw output) directly;
    the normalised tensor is only materialised for the ones that go through MIOpen."""
    n, h, w, cin = x.shape
    f0 = self._fusable(p + 'conv_0', h, w, stride)
    fp = use_projection and self._fusable(p + 'proj_conv', h, w, stride)
    strided = stride == 2
    if not f0 or (use_projection and not fp):
      y, ysub = self._hip_norm_relu(x, st, p + 'bn_0', tag + 'a', pad=strided, sub=strided)
    shortcut = x
    if (use_projection and f0 and fp and self.fuse_proj and not self._small_now and p in self._wdual and self._wdual[p][3] == stride
        and self._fusable(p + 'conv_1', -(-h // stride), -(-w // stride), 1)):
      y0, shortcut, st0 = self._fused_conv_dual(x, st, p, tag, stride, next_norm=p + 'bn_1')
      return self._fused_conv(y0, st0, p + 'bn_1', p + 'conv_1', shortcut, tag + f'r{parity}', next_norm=next_norm)
    if use_projection:
      if fp:
        shortcut, _ = self._fused_conv(x, st, p + 'bn_0', p + 'proj_conv', None, tag + 'p', stride, stats=False)
      else:
        shortcut = self._hip_conv(ysub if strided else y, p + 'proj_conv')
    if f0:
      y0, st0 = self._fused_conv(x, st, p + 'bn_0', p + 'conv_0', None, tag + 'c', stride,
                                 reuse_ss=bool(use_projection and fp) and not self._small_now,   # (proj_conv just merged bn_0's pairs into `ss`;
                                 #  the few-frame form merges inside every consuming launch and writes no `ss`)
                                 next_norm=p + 'bn_1')
    else:
      y0 = self._hip_conv(y, p + 'conv_0', stride, 0 if strided else 1)
      st0 = self._hip_stats(y0)
    if self._fusable(p + 'conv_1', y0.shape[1], y0.shape[2], 1):
      # the result becomes the next block's residual stream: it may not alias this block's (the shortcut)
      return self._fused_conv(y0, st0, p + 'bn_1', p + 'conv_1', shortcut, tag + f'r{parity}', next_norm=next_norm)
    y, _ = self._hip_norm_relu(y0, st0, p + 'bn_1', tag + 'b')
    y1 = self._hip_conv(y, p + 'conv_1', 1, 1)
    return y1, self._hip_stats(y1, shortcut)   # y1 += shortcut, fused with the next norm's statistics

  def _hip_l2norm(self, x_nhwc, out=None, op=None, tiled=None):
    """op / tiled (bf16 build): the normalised map once more in the hot path's operand type -- row-major bf16 and
    (256 channels) the cost-volume kernel's tile order -- so that the hot path does not re-read the f32 grids to cast
    them (tapir_l2_normalize_staged)."""
    lib, ctx = self.engine
    n, h, w, c = x_nhwc.shape
    if out is None:
      out = torch.empty((n, h, w, c), dtype=torch.float32, device=self.device)
    assert out.shape == (n, h, w, c) and out.is_contiguous() and out.dtype == torch.float32
    if op is None:
      self._check(lib.tapir_l2_normalize(ctx, x_nhwc.data_ptr(), out.data_ptr(), n * h * w, c,
                                         self._stream()), 'tapir_l2_normalize')
    else:
      assert op.is_contiguous() and op.dtype == torch.bfloat16 and op.numel() == out.numel()
      assert tiled is None or (tiled.is_contiguous() and tiled.shape[0] == n)
      self._check(lib.tapir_l2_normalize_staged(ctx, x_nhwc.data_ptr(), out.data_ptr(), op.data_ptr(),
                                                tiled.data_ptr() if tiled is not None else None, n * h * w, c, h * w,
                                                self._stream()), 'tapir_l2_normalize_staged')
    return out

  @staticmethod
  def staged_like(low, hi):
    """bf16 companions of a pair of f32 grids for _hip_l2norm: (low row-major, low in tile order, hi row-major).  The
    tile-order buffer is zeroed once: cells past the end of a frame's last tile of 16 are never written."""
    n, h, w, c = low.shape
    return (torch.empty_like(low, dtype=torch.bfloat16),
            torch.zeros((n, ((h * w + 15) // 16) * 16 * c), dtype=torch.bfloat16, device=low.device),
            torch.empty_like(hi, dtype=torch.bfloat16))

  def _features_hip(self, frames_nhwc, out_low=None, out_hi=None, staged=None):
    st = None
    if self._hip_now and 'stem' in self.hip_convs and self._stem_ws is not None:
      import ctypes
      lib, ctx = self.engine
      n, h, w, _ = frames_nhwc.shape
      rows, tiles = ctypes.c_int(), ctypes.c_int()
      if (frames_nhwc.dtype == torch.float32 and
          lib.tapir_stem_plan(ctx, h, w, ctypes.byref(rows), ctypes.byref(tiles)) == 0):
        ho, wo = -(-h // 2), -(-w // 2)
        fr = frames_nhwc.contiguous()
        x = self._buf(('stem', n, ho, wo), (n, ho, wo, 64), self.dtype)
        part = self._buf(('stempart', n, tiles.value), (n, tiles.value, 64, 2), torch.float32)
        first = 'resnet_torch.block_groups.0.blocks.0.bn_0'
        nn, ssn = self._next_norm(first, n, 64, 'stem')
        self._check(lib.tapir_stem_conv_nn(ctx, fr.data_ptr(), self._stem_ws, x.data_ptr(), part.data_ptr(), n, h, w,
                                           ctypes.byref(nn) if nn is not None else None, self._stream()),
                    'tapir_stem_conv_nn')
        st = _Stats(part, tiles.value, rows.value * wo, ssn, first if nn is not None else None)
    if st is None:
      x = frames_nhwc.to(self.dtype).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
      w0 = self.w['resnet_torch.initial_conv.weight']
      x = F.conv2d(_same_pad(x, w0.shape[-1], 2), w0, None, stride=2).permute(0, 2, 3, 1).contiguous()
      st = self._hip_stats(x)
    strides = (1, 2, 2, 1)
    unit1 = None
    blocks = [(g, b) for g in range(4) for b in range(self.blocks_per_group[g])]
    for i, (g, b) in enumerate(blocks):
      nxt = 'resnet_torch.block_groups.%d.blocks.%d.bn_0' % blocks[i + 1] if i + 1 < len(blocks) else None
      x, st = self._hip_block(x, st, f'resnet_torch.block_groups.{g}.blocks.{b}.',
                              strides[g] if b == 0 else 1, b == 0, f'g{g}', b & 1, next_norm=nxt)
      if g == 1 and b == self.blocks_per_group[g] - 1:
        unit1 = x
    if self.extra_convs:
      # (like the ResNet convolutions: a single frame -- the online model -- gives the HIP kernels 16-64
      # workgroups per launch; the library's split-K kernels win there)
      # (few-frame clips: the 256 -> 1024 convolution takes the few-frame form inside tapir_xconv_nt, csrc/conv_small.hpp)
      xe = self._extra_convs_hip(x) if (self.extra_convs_mode == 'hip' and self._hip_now) else None
      x = xe if xe is not None else self._extra_convs(x.permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()
    if staged is not None:
      # the caller publishes these buffers to the hot path (last_staged -> tapir_set_staged_grid): they must be written
      if x.shape[-1] != 256:
        raise RuntimeError('tapnet_amd.backbone: staged grid copies requested for a %d-channel low-res map '
                           '(_stage_ok and the weights disagree)' % x.shape[-1])
      return self._hip_l2norm(x, out_low, staged[0], staged[1]), self._hip_l2norm(unit1, out_hi, staged[2])
    return self._hip_l2norm(x, out_low), self._hip_l2norm(unit1, out_hi)

  # -- public ---------------------------------------------------------------
  @torch.no_grad()
  def features(self, frames_nhwc: torch.Tensor, chunk: Optional[int] = None, borrow: bool = False,
               global_frames: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """frames [N,H,W,3] f32 in [-1,1] -> (lowres [N,H/8,W/8,256], hires [N,H/4,W/4,128]) f32,
    L2-normalised, contiguous channels-last.  borrow=True: the caller consumes the grids before the
    next call with this shape and may get the graph's own output buffers (no 150-MB copy per clip).
    chunk (feature_extractor_chunk_size, tapir_model.py:689-703): frames per backbone pass -- bounds the
    scratch memory, so such calls are launched eagerly (a captured graph keeps full-clip static buffers).
    global_frames: frame count of the whole clip when this call sees one rank's shard of it."""
    n, H, W = frames_nhwc.shape[:3]
    self.last_staged = None   # borrow=True, bf16 engine: (low16, low_tiled, hi16) written next to the returned f32 grids
    self._clip_frames = max(n, int(global_frames or 0))
    self._hip_now = self._use_hip_convs(self._clip_frames)
    small = self._hip_now and self._small_ok(self._clip_frames)
    if self.engine is not None and (small or self._small_now):
      self._check(self.engine[0].tapir_conv_set_small(self.engine[1], int(small)), 'tapir_conv_set_small')
    self._small_now = small
    half = lambda v: -(-v // 2)
    last = lambda g: f'resnet_torch.block_groups.{g}.blocks.{self.blocks_per_group[g] - 1}.conv_1.weight'
    c_low = self.w[last(3)].shape[0]
    if self.extra_convs:
      c_low = self.w['extra_convs.blocks.4.conv_1.weight'].shape[0]
    c_hi = self.w[last(1)].shape[0]
    low = torch.empty((n, half(half(half(H))), half(half(half(W))), c_low), dtype=torch.float32, device=self.device)
    hi = torch.empty((n, half(half(H)), half(half(W)), c_hi), dtype=torch.float32, device=self.device)
    if n == 0:
      return low, hi
    # chunked calls exist to BOUND the scratch (feature_extractor_chunk_size, tapir_model.py:689-703): one stream,
    # one chunk's scratch alive at a time (every stream lane keeps its own buffers)
    streams = 1 if chunk else max(1, min(int(self.streams), n // 8))
    # hipGraph replay: a clip's backbone is ~60 launches of 10-100 us kernels, which one Python thread
    # cannot issue as fast as the GPU retires them (2.04 ms wall against 1.7 ms of kernels for 48 frames).
    # From the third call with the same shape on, the launches are replayed from a captured graph.
    if chunk:
      bounds = [(s, min(s + chunk, n)) for s in range(0, n, chunk)]
    else:
      # frame groups: one per stream, or more (TAPIR_BACKBONE_GROUPS: the groups of a stream run one after the other,
      # which shrinks the set of activations alive at any time -- tools, A/B)
      groups = max(streams, int(os.environ.get('TAPIR_BACKBONE_GROUPS', '0') or 0))
      per = -(-n // groups)
      bounds = [(s, min(s + per, n)) for s in range(0, n, per)]
    key = (n, self._clip_frames, H, W, self._hip_now, tuple(sorted(self.hip_convs)), self.extra_convs_mode, streams, tuple(bounds),
           bool(self.fuse_proj), bool(self.fuse_finalize), self.conv_mode, self._small_now)   # (everything that changes WHICH kernels a pass launches)
    if chunk or os.environ.get('TAPIR_BACKBONE_GRAPH', '1') == '0':   # (chunked: see above; profilers that need
      key = None                                                       #  every dispatch on its own)
    if (key is not None and self.graph_min_frames and n >= self.graph_min_frames
        and not torch.cuda.is_current_stream_capturing()):
      ent = self._graphs.pop(key, None)            # (re-inserted below: the dict is kept in LRU order)
      if ent is None:
        ent = {'seen': 0}
      self._graphs[key] = ent
      ent['seen'] += 1
      # entries that were only counted cost nothing and are trimmed here; captured graphs (large static buffers) are
      # evicted only at the moment another one is about to be captured (below): a one-off shape that never reaches
      # its third call does not push a captured graph out
      if len(self._graphs) > 64:
        for k in [k for k, e in self._graphs.items() if 'graph' not in e and k != key][:32]:
          self._graphs.pop(k)
      if 'graph' not in ent and not ent.get('failed') and ent['seen'] >= 3:
        # at most four captured graphs resident, this one included: drop the least recently used one
        captured = [k for k, e in self._graphs.items() if 'graph' in e and k != key]
        if len(captured) >= 4:
          self._graphs.pop(captured[0])
        ent['in'] = frames_nhwc.contiguous().clone()
        ent['low'], ent['hi'] = torch.empty_like(low), torch.empty_like(hi)
        ent['staged'] = self.staged_like(low, hi) if self._stage_ok(low) else None
        torch.cuda.synchronize(self.device)
        try:
          g = torch.cuda.CUDAGraph()
          with torch.cuda.graph(g):   # (the side streams fork from and join the capturing stream)
            self._run_groups(ent['in'], ent['low'], ent['hi'], bounds, streams, ent['staged'])
          ent['graph'] = g
        except RuntimeError as e:     # e.g. another thread touched the device during the capture
          import warnings
          warnings.warn(f'tapnet_amd.backbone: hipGraph capture failed ({e}); launching eagerly')
          ent['failed'] = True
          for k in ('in', 'low', 'hi', 'staged'):
            ent.pop(k, None)
          torch.cuda.synchronize(self.device)
      if 'graph' in ent:
        ent['in'].copy_(frames_nhwc)
        ent['graph'].replay()
        if borrow:
          self.last_staged = ent['staged']
          return ent['low'], ent['hi']
        low.copy_(ent['low'])
        hi.copy_(ent['hi'])
        return low, hi
    staged = self.staged_like(low, hi) if (borrow and self._stage_ok(low)) else None
    self._run_groups(frames_nhwc, low, hi, bounds, streams, staged)
    self.last_staged = staged
    return low, hi

  def _stage_ok(self, low):
    """operand-type copies from the L2-normalise kernel: bf16 engine, HIP path, 256-channel low-res map"""
    return (self.dtype == torch.bfloat16 and self.engine is not None and low.shape[-1] == 256
            and os.environ.get('TAPIR_STAGE_GRIDS', '1') != '0')

  def _run_groups(self, frames_nhwc, low, hi, bounds, streams, staged=None):
    cur = torch.cuda.current_stream(self.device)
    while len(self._side_streams) < streams - 1:
      self._side_streams.append(torch.cuda.Stream(self.device))
    saved = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = bool(self.miopen_exhaustive_search)
    try:
      if streams > 1:
        fork = cur.record_event()
      for i, (s, e) in enumerate(bounds):
        lane = i % streams
        st = cur if lane == 0 else self._side_streams[lane - 1]
        if lane and i < streams:
          st.wait_event(fork)
        self._lane = lane
        with torch.cuda.stream(st):   # (issuing / capturing the groups' launches block by block round-robin instead of group after
          #  group measures the same: profiles/r06_ab_interleave.txt)
          self._features_hip(frames_nhwc[s:e], low[s:e], hi[s:e],
                             None if staged is None else tuple(t[s:e] for t in staged))
      for st in self._side_streams[:streams - 1]:
        cur.wait_stream(st)
    finally:
      self._lane = 0
      torch.backends.cudnn.benchmark = saved


def resize_bilinear(video: torch.Tensor, resolution: Tuple[int, int], antialias: bool = False) -> torch.Tensor:
  """[B,T,H,W,3] -> [B,T,h,w,3].  The reference's torch twin uses
  F.interpolate(bilinear, align_corners=False) (tapnet/torch/utils.py:26-42) -- the default here,
  and what the fixtures pin.  The JAX model's jax.image.resize(method='bilinear')
  (tapir_model.py:670) additionally anti-aliases when DOWN-sampling (triangle kernel widened by the
  scale factor); `antialias=True` selects that behaviour through F.interpolate(antialias=True), the
  same filter family.  Identical for up-sampling and for the no-op 256->256 case; it matters for
  inputs larger than initial_resolution (e.g. 512 -> 256, BASELINE configs[4]).  Opt-in for torch-named
  weights (TAPIR(..., jax_antialias_resize=True)), the default when ParameterizedTAPIR is given a Haiku tree;
  held to a restatement of jax/_src/image/scale.py and, through it, to the JAX text run over numpy stand-ins
  (tests/test_jax_reference_pin.py::test_gpu_matches_the_jax_text[bootstapir_multires])."""
  b, t, h, w, c = video.shape
  if (int(resolution[0]), int(resolution[1])) == (h, w) and os.environ.get('TAPIR_IDENTITY_RESIZE', '0') != '1':
    # (TAPIR_IDENTITY_RESIZE=1: run the resize anyway -- the A/B switch of profiles/r04_ab_latency_chains.txt)
    # The reference resizes even to the size the video already has (the quirk at tapir_model.py:667 makes the first
    # level always take this branch).  At equal size align_corners=False samples every pixel centre with weights
    # exactly (1, 0): the identity, bit for bit, with or without the anti-aliasing filter (torch's kernels copy) --
    # tests/test_host_logic.py::test_resize_to_the_same_size_is_the_identity.  Returning the input saves the
    # benchmarked 256 x 256 clip two transposing passes and a copy over its 38 MB (~65 us of a 4.7 ms step).
    return video
  x = video.permute(0, 1, 4, 2, 3).reshape(b, t * c, h, w)
  down = resolution[0] < h or resolution[1] < w
  x = F.interpolate(x, size=tuple(resolution), mode='bilinear', align_corners=False,
                    antialias=bool(antialias and down))
  return x.reshape(b, t, c, resolution[0], resolution[1]).permute(0, 1, 3, 4, 2)
