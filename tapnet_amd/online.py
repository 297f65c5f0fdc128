"""Per-frame (online / causal) tracking: the loop of ``tapnet/live_demo.py:51-77`` and
``tapnet/pytorch_live_demo.py:44-85`` as a session object whose whole per-frame step --
backbone on one frame, cost volume, ``num_pips_iter`` causal refinement iterations, causal
state hand-over -- is replayed from a captured hipGraph.

The reference's loop is

    query_features = online_model_init(first_frame, points)           # get_query_features
    causal_state   = tapir.construct_initial_causal_state(N, len(query_features.resolutions) - 1)
    for frame in stream:
      prediction, causal_state = online_model_predict(frame, query_features, causal_state)

With T = 1 the step is ~250 kernel launches of a few microseconds each: launch-bound.  Here the
launches are recorded once (``torch.cuda.CUDAGraph``, i.e. hipStreamBeginCapture / hipGraphLaunch on
ROCm; the HIP kernels are enqueued on the capturing stream through the C ABI, which does no
allocation or host synchronisation after ``tapir_reserve``) and replayed per frame.  The causal
state lives in two packed device buffers [iters, blocks, N, 2, 512 | 2048] that alternate as
input / output (two graphs, one per direction), so no state is copied between frames.
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from tapnet_amd import _ffi, model_utils
from tapnet_amd.tapir_model import TAPIR, FeatureGrids, QueryFeatures, _res_hw


class OnlineTracker:
  """One tracked stream: fixed frame size and point capacity."""

  def __init__(self, model: TAPIR, num_points: int, frame_hw: Tuple[int, int], use_graph: bool = True):
    if not model.use_causal_conv:
      raise ValueError('online tracking needs a causal model (use_causal_conv=True)')
    self.model = model
    self.n = int(num_points)
    self.hw = (int(frame_hw[0]), int(frame_hw[1]))
    self.use_graph = use_graph
    dev = model.device
    # refinement levels of this frame size, fixed for the session (the reference sizes the state the
    # same way: len(query_features.resolutions) - 1 levels x num_pips_iter iterations,
    # tapir_clustering.py:819-820, live_demo.py:120): one level when the frame has
    # initial_resolution, more for larger frames (model_utils.generate_default_resolutions)
    self.refinement_resolutions = [tuple(int(v) for v in r) for r in
                                   model_utils.generate_default_resolutions(self.hw, model.initial_resolution)]
    self.nl = 1 + len(self.refinement_resolutions)
    self.ni = model.num_pips_iter * (self.nl - 1)
    nb = model.num_mixer_blocks
    self._frame = torch.zeros((1, 1, self.hw[0], self.hw[1], 3), device=dev)
    self._state = [(torch.zeros((self.ni, nb, self.n, 2, 512), device=dev),
                    torch.zeros((self.ni, nb, self.n, 2, 2048), device=dev)) for _ in range(2)]
    self._cur = 0                     # index of the state buffer that holds the current state
    self._out = dict(tracks=torch.zeros((self.ni + 1, 1, self.n, 1, 2), device=dev),
                     occlusion=torch.zeros((self.ni + 1, 1, self.n, 1), device=dev),
                     expected_dist=torch.zeros((self.ni + 1, 1, self.n, 1), device=dev))
    self._qf: Optional[QueryFeatures] = None
    self._graphs = [None, None]
    self._warm = False
    self._pinned = False

  def close(self) -> None:
    """Drops the captured graphs and releases this session's pin on the model's workspaces."""
    self._graphs = [None, None]
    self._warm = False
    if getattr(self, '_pinned', False):   # (__del__ also runs when __init__ raised half-way)
      self._pinned = False
      try:
        self.model._lib.tapir_pin_workspaces(self.model._ctx, 0)
      except Exception:   # interpreter shutdown / model already destroyed
        pass

  def __del__(self):
    self.close()

  def check(self) -> None:
    """Raises if the mixer's persistent launch (csrc/mixer_online.hpp: up to 256 points) gave up waiting for one of its
    workgroups -- the device could not hold its 256 workgroups at once; the tracks of that frame are NaN.  Synchronises the
    device: call it where the results are read anyway, not per frame in a latency-critical loop."""
    import ctypes
    word = ctypes.c_uint(0)
    self.model._check(self.model._lib.tapir_online_sync_error(self.model._ctx, ctypes.byref(word)), 'tapir_online_sync_error')
    if word.value:
      raise RuntimeError(f'online mixer: a workgroup timed out at a cluster barrier (word {word.value:#x}); rerun with '
                         'TAPIR_SMALL_GEMM=2 (separate launches) if the GPU is shared')

  # ------------------------------------------------------------------ points
  def init(self, frames, query_points) -> QueryFeatures:
    """live_demo.online_model_init: query features of `query_points` [1,N,3] (t,y,x) in `frames`
    [1,T,H,W,3] (typically the first frame).  Resets the causal state."""
    m = self.model
    fg = m.get_feature_grids(m._dev(frames), False, self.refinement_resolutions)
    qf = m.get_query_features(m._dev(frames), False, m._dev(query_points), fg)
    return self.set_query_features(qf)

  def set_query_features(self, qf: QueryFeatures) -> QueryFeatures:
    """Track the points whose query features are `qf` (e.g. extracted from other frames or other
    videos, tapir_clustering.py:1133-1140) from a zero causal state."""
    m = self.model
    qf = QueryFeatures(tuple(m._dev(t) for t in qf.lowres), tuple(m._dev(t) for t in qf.hires),
                       qf.resolutions)
    if qf.lowres[0].shape[1] != self.n:
      raise ValueError(f'expected {self.n} query points')
    if len(qf.lowres) != self.nl:
      raise ValueError(f'query features hold {len(qf.lowres)} levels, the session {self.nl} '
                       f'(frame {self.hw}, initial resolution {self.model.initial_resolution})')
    if self._qf is None:
      self._qf = QueryFeatures(tuple(t.clone() for t in qf.lowres), tuple(t.clone() for t in qf.hires),
                               qf.resolutions)
    else:   # keep the buffers the graphs point at
      for d, s in zip(self._qf.lowres + self._qf.hires, qf.lowres + qf.hires):
        d.copy_(s)
    for a, b in self._state:
      a.zero_(); b.zero_()
    return self._qf

  def update_points(self, idx: Sequence[int], frames, query_points) -> None:
    """TAPIR.update_query_features (tapir_model.py:1172-1203), in place: new query features and a
    zero causal state for the points `idx`."""
    m = self.model
    fg = m.get_feature_grids(m._dev(frames), False, self.refinement_resolutions)
    new = m.get_query_features(m._dev(frames), False, m._dev(query_points), fg)
    ix = torch.as_tensor(np.asarray(idx), device=m.device, dtype=torch.long)
    for d, s in zip(self._qf.lowres + self._qf.hires, new.lowres + new.hires):
      d[:, ix] = s
    a, b = self._state[self._cur]
    a[:, :, ix] = 0.0
    b[:, :, ix] = 0.0

  # ------------------------------------------------------------------ step
  def _run(self, src: int) -> None:
    """backbone(frame) + estimate_trajectories with state src -> 1 - src, outputs into self._out"""
    m = self.model
    fg = m.get_feature_grids(self._frame, False, self.refinement_resolutions)
    nl = len(fg.lowres)
    assert nl == self.nl   # the state / output buffers are sized for exactly these levels
    a = _ffi.TapirTrajArgs()
    a.B, a.Q, a.T, a.n_levels = 1, self.n, 1, nl
    keep = []
    for l in range(nl):
      lo, hi = m._dev(fg.lowres[l]), m._dev(fg.hires[l])
      keep += [lo, hi]
      a.lowres[l] = lo.data_ptr(); a.hires[l] = hi.data_ptr()
      a.lowres_h[l], a.lowres_w[l] = lo.shape[2:4]
      a.hires_h[l], a.hires_w[l] = hi.shape[2:4]
      a.res_h[l], a.res_w[l] = _res_hw(fg.resolutions[l])
      a.q_lowres[l] = self._qf.lowres[l].data_ptr(); a.q_hires[l] = self._qf.hires[l].data_ptr()
    a.video_h, a.video_w = self.hw
    a.ctx1_in, a.ctx2_in = self._state[src][0].data_ptr(), self._state[src][1].data_ptr()
    a.ctx1_out, a.ctx2_out = self._state[1 - src][0].data_ptr(), self._state[1 - src][1].data_ptr()
    a.tracks = self._out['tracks'].data_ptr()
    a.occlusion = self._out['occlusion'].data_ptr()
    a.expected_dist = self._out['expected_dist'].data_ptr()
    m._check(m._lib.tapir_estimate_trajectories(m._ctx, ctypes.byref(a), m._stream()),
             'tapir_estimate_trajectories')
    self._keep = keep

  def _prepare(self) -> None:
    """workspaces, MIOpen solver search and scratch buffers: everything that allocates"""
    m = self.model
    big = max([tuple(m.initial_resolution)] + self.refinement_resolutions, key=lambda r: r[0] * r[1])
    m.reserve(1, self.n, 1, (big[0] // 8, big[1] // 8))
    saved = [(a.clone(), b.clone()) for a, b in self._state]
    for _ in range(2):
      self._run(0)
    torch.cuda.synchronize(m.device)
    for (a, b), (sa, sb) in zip(self._state, saved):
      a.copy_(sa); b.copy_(sb)
    if self.use_graph:
      # the graphs bake in the engine's workspace pointers: from here on a larger call on this
      # model gets an error instead of reallocating them under the graphs (include/tapir_hip.h)
      m._check(m._lib.tapir_pin_workspaces(m._ctx, 1), 'tapir_pin_workspaces')
      self._pinned = True
      try:
        for src in (0, 1):
          g = torch.cuda.CUDAGraph()
          with torch.cuda.graph(g):
            self._run(src)
          self._graphs[src] = g
      except Exception:
        # a failed capture must not leave the model pinned: every later, larger call on the shared model
        # would fail with 'workspace growth while pinned'
        self.close()
        raise
      torch.cuda.synchronize(m.device)
      for (a, b), (sa, sb) in zip(self._state, saved):   # the capture itself does not execute,
        a.copy_(sa); b.copy_(sb)                        # but keep the state exactly as it was
    self._warm = True

  @torch.no_grad()
  def step(self, frame) -> Dict[str, Any]:
    """live_demo.online_model_predict: `frame` [1,1,H,W,3] (or [H,W,3]) in [-1,1] -> the last
    refinement iteration's tracks [1,N,1,2] (x,y), occlusion and expected_dist logits [1,N,1].
    The returned tensors are views of buffers that the next step overwrites."""
    if self._qf is None:
      raise RuntimeError('call init() first')
    f = self.model._dev(frame)
    self._frame.copy_(f.reshape(self._frame.shape))
    if not self._warm:
      self._prepare()
    if self.use_graph:
      self._graphs[self._cur].replay()
    else:
      self._run(self._cur)
    self._cur = 1 - self._cur
    return {k: v[-1] for k, v in self._out.items()}

  @property
  def causal_state(self):
    """the current state in the reference's structure (list of dicts), for interoperability"""
    return self.model._unpack_state(self._state[self._cur], 1, self.n)
