"""TEST INFRASTRUCTURE ONLY: drives tests/hipemu/libtapir_emu.so (the HIP sources
compiled for the host against the fiber emulator) through the same C ABI as the
gfx950 library, with numpy arrays standing in for device memory.  Never
imported by tapnet_amd/."""
import ctypes
import os
import subprocess

import numpy as np

from tapnet_amd import _ffi

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, 'hipemu')
EMU_LIB = os.environ.get('TAPIR_EMU_LIB') or os.path.join(EMU_DIR, 'libtapir_emu.so')   # (tests/hipemu/run_asan.sh: the sanitizer build)
CSRC = os.path.join(os.path.dirname(HERE), 'tapnet_amd', 'csrc')


def build_emu(force=False):
  srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.hpp'))]
  srcs += [os.path.join(EMU_DIR, 'hip', 'hip_runtime.h'), os.path.join(EMU_DIR, 'emu_switch.cpp'),
           os.path.join(os.path.dirname(HERE), 'include', 'tapir_hip.h')]
  if (not force and os.path.exists(EMU_LIB)
      and os.path.getmtime(EMU_LIB) >= max(os.path.getmtime(s) for s in srcs)):
    return EMU_LIB
  subprocess.check_call([os.path.join(EMU_DIR, 'build_emu.sh')])
  return EMU_LIB


_lib = None


def emu_lib():
  global _lib
  if _lib is None:
    _lib = _ffi.declare_prototypes(ctypes.CDLL(build_emu()))
  return _lib


def _p(a):
  return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def f32(a):
  return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


class EmuEngine:
  def __init__(self, weights, pyramid_level=1, num_pips_iter=4, num_mixer_blocks=12,
               use_causal_conv=False, softmax_temperature=20.0, initial_resolution=(256, 256),
               dtype=_ffi.TAPIR_F32):
    self.lib = emu_lib()
    self.cfg = _ffi.TapirCfg(pyramid_level, num_pips_iter, num_mixer_blocks,
                             int(use_causal_conv), softmax_temperature,
                             initial_resolution[0], initial_resolution[1], dtype)
    self.ctx = ctypes.c_void_p()
    self._chk(self.lib.tapir_create(ctypes.byref(self.ctx), ctypes.byref(self.cfg), 0), 'create')
    for k, v in weights.items():
      v = f32(v)
      shape = (ctypes.c_int64 * v.ndim)(*v.shape)
      self._chk(self.lib.tapir_set_weight(self.ctx, k.encode(), _p(v), shape, v.ndim), 'set_weight')
    self._chk(self.lib.tapir_finalize_weights(self.ctx), 'finalize')
    self.nb = num_mixer_blocks
    self.P = num_pips_iter

  def _chk(self, rc, what):
    _ffi.check(self.lib, self.ctx, rc, what)

  def close(self):
    if self.ctx:
      self.lib.tapir_destroy(self.ctx)
      self.ctx = ctypes.c_void_p()

  def build_cost_volume(self, qfeat, grid):
    qfeat, grid = f32(qfeat), f32(grid)
    B, Q, C = qfeat.shape
    _, T, h, w, _ = grid.shape
    out = np.zeros((B, Q, T, h, w), np.float32)
    self._chk(self.lib.tapir_build_cost_volume(self.ctx, _p(qfeat), _p(grid), B, Q, T, h, w, C,
                                               _p(out), None), 'build_cost_volume')
    return out

  def tracks_from_cost_volume(self, qfeat, grid, qpts):
    qfeat, grid, qpts = f32(qfeat), f32(grid), f32(qpts)
    B, Q, C = qfeat.shape
    _, T, h, w, _ = grid.shape
    pts = np.zeros((B, Q, T, 2), np.float32)
    occ = np.zeros((B, Q, T), np.float32)
    expd = np.zeros((B, Q, T), np.float32)
    self._chk(self.lib.tapir_tracks_from_cost_volume(
        self.ctx, _p(qfeat), _p(grid), _p(qpts), B, Q, T, h, w, _p(pts), _p(occ), _p(expd), None),
        'tracks_from_cost_volume')
    return pts, occ, expd

  def tapnet_tracks_from_cost_volume(self, qfeat, grid, qpts):
    qfeat, grid, qpts = f32(qfeat), f32(grid), f32(qpts)
    B, Q, _ = qfeat.shape
    _, T, h, w, _ = grid.shape
    pts = np.zeros((B, Q, T, 2), np.float32); occ = np.zeros((B, Q, T), np.float32)
    self._chk(self.lib.tapir_tapnet_tracks_from_cost_volume(self.ctx, _p(qfeat), _p(grid), _p(qpts), B, Q, T, h, w,
                                                            _p(pts), _p(occ), None), 'tapnet head')
    return pts, occ

  def cycle_consistency_tracks(self, qfeat, grid, qpts, im_hw, temperature=10.0, threshold=48.0):
    qfeat, grid, qpts = f32(qfeat), f32(grid), f32(qpts)
    B, Q, _ = qfeat.shape
    _, T, h, w, _ = grid.shape
    pts = np.zeros((B, Q, T, 2), np.float32); occ = np.zeros((B, Q, T), np.float32); inv = np.zeros((B, Q, T, 2), np.float32)
    self._chk(self.lib.tapir_cycle_consistency_tracks(self.ctx, _p(qfeat), _p(grid), _p(qpts), B, Q, T, h, w, int(im_hw[0]),
                                                      int(im_hw[1]), temperature, threshold, _p(pts), _p(occ), _p(inv), None),
              'cycle consistency')
    return pts, occ, inv

  def get_query_features(self, grid, qpts, video_hw):
    grid, qpts = f32(grid), f32(qpts)
    B, T, h, w, C = grid.shape
    Q = qpts.shape[1]
    out = np.zeros((B, Q, C), np.float32)
    self._chk(self.lib.tapir_get_query_features(self.ctx, _p(grid), _p(qpts), B, Q, T, h, w, C,
                                                video_hw[0], video_hw[1], _p(out), None), 'qf')
    return out

  def pips_mixer(self, x, ctx1=None, ctx2=None, get_ctx=False):
    x = f32(x)
    N, T, _ = x.shape
    out = np.zeros((N, T, 388), np.float32)
    c1o = np.zeros((self.nb, N, 2, 512), np.float32) if get_ctx else None
    c2o = np.zeros((self.nb, N, 2, 2048), np.float32) if get_ctx else None
    self._chk(self.lib.tapir_pips_mixer(self.ctx, _p(x), N, T, _p(out), _p(f32(ctx1)),
                                        _p(f32(ctx2)), _p(c1o), _p(c2o), None), 'pips_mixer')
    return (out, c1o, c2o) if get_ctx else out

  def refine_pips(self, queries, pyramid, pos, occ, expd, last_iter, orig_hw, resize_hw):
    queries = [f32(q) for q in queries]
    pyramid = [f32(g) for g in pyramid]
    pos, occ, expd, last_iter = f32(pos), f32(occ), f32(expd), f32(last_iter)
    B, Q, T, _ = pos.shape
    pyr = _ffi.TapirPyramid()
    pyr.n_levels = len(pyramid)
    for l, (q, g) in enumerate(zip(queries, pyramid)):
      pyr.query[l] = q.ctypes.data
      pyr.grid[l] = g.ctypes.data
      pyr.h[l], pyr.w[l], pyr.C[l] = g.shape[2], g.shape[3], g.shape[4]
    po = np.zeros_like(pos); oo = np.zeros_like(occ); eo = np.zeros_like(expd)
    fo = np.zeros((B, Q, T, 384), np.float32)
    self._chk(self.lib.tapir_refine_pips(
        self.ctx, ctypes.byref(pyr), B, Q, T, _p(pos), _p(occ), _p(expd), _p(last_iter),
        orig_hw[0], orig_hw[1], resize_hw[0], resize_hw[1], _p(po), _p(oo), _p(eo), _p(fo),
        None, None, None, None, None), 'refine_pips')
    return po, oo, eo, fo

  def estimate_trajectories(self, video_hw, lowres, hires, resolutions, q_lowres, q_hires,
                            query_points, ctx_in=None, get_ctx=False):
    lowres = [f32(x) for x in lowres]; hires = [f32(x) for x in hires]
    q_lowres = [f32(x) for x in q_lowres]; q_hires = [f32(x) for x in q_hires]
    query_points = f32(query_points)
    B, T = lowres[0].shape[:2]
    Q = q_lowres[0].shape[1]
    nl = len(lowres)
    ni = self.P * (nl - 1)
    a = _ffi.TapirTrajArgs()
    a.B, a.Q, a.T, a.n_levels = B, Q, T, nl
    for l in range(nl):
      a.lowres[l] = lowres[l].ctypes.data; a.hires[l] = hires[l].ctypes.data
      a.lowres_h[l], a.lowres_w[l] = lowres[l].shape[2:4]
      a.hires_h[l], a.hires_w[l] = hires[l].shape[2:4]
      a.res_h[l], a.res_w[l] = resolutions[l]
      a.q_lowres[l] = q_lowres[l].ctypes.data; a.q_hires[l] = q_hires[l].ctypes.data
    a.query_points = query_points.ctypes.data if query_points is not None else None
    a.video_h, a.video_w = video_hw
    tr = np.zeros((ni + 1, B, Q, T, 2), np.float32)
    oc = np.zeros((ni + 1, B, Q, T), np.float32)
    ex = np.zeros((ni + 1, B, Q, T), np.float32)
    a.tracks, a.occlusion, a.expected_dist = tr.ctypes.data, oc.ctypes.data, ex.ctypes.data
    keep = []
    if ctx_in is not None:
      c1, c2 = f32(ctx_in[0]), f32(ctx_in[1]); keep += [c1, c2]
      a.ctx1_in, a.ctx2_in = c1.ctypes.data, c2.ctypes.data
    c1o = c2o = None
    if get_ctx:
      c1o = np.zeros((ni, self.nb, B * Q, 2, 512), np.float32)
      c2o = np.zeros((ni, self.nb, B * Q, 2, 2048), np.float32)
      a.ctx1_out, a.ctx2_out = c1o.ctypes.data, c2o.ctypes.data
    self._chk(self.lib.tapir_estimate_trajectories(self.ctx, ctypes.byref(a), None), 'estimate')
    out = dict(tracks=tr, occlusion=oc, expected_dist=ex)
    if get_ctx:
      out['ctx'] = (c1o, c2o)
    return out
