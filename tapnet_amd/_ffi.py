"""ctypes binding of include/tapir_hip.h (the C ABI of libtapir_hip.so).

north_star asks for a "thin C-ABI cffi layer"; cffi is not installed in this
image, ctypes (stdlib) binds the same C ABI.  The product path loads ONLY the
in-tree gfx950 library ``tapnet_amd/csrc/libtapir_hip.so`` and raises if it is
missing -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_long, c_void_p

TAPIR_OK = 0
TAPIR_ERR_INVALID = -1
TAPIR_ERR_HIP = -2
TAPIR_ERR_UNSUPPORTED = -3
TAPIR_ERR_WEIGHTS = -4
TAPIR_F32 = 0
TAPIR_BF16 = 1
TAPIR_MAX_LEVELS = 8

# (TAPIR_HIP_LIB: tools/kbench.py points this at a -DTAPIR_EXPERIMENTS build of the same sources --
# still a gfx950 library built from tapnet_amd/csrc, never a fallback)
LIB_PATH = os.environ.get('TAPIR_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                           'csrc', 'libtapir_hip.so')

c_float_p = POINTER(c_float)


class TapirCfg(ctypes.Structure):
  _fields_ = [('pyramid_level', c_int), ('num_pips_iter', c_int), ('num_mixer_blocks', c_int),
              ('use_causal_conv', c_int), ('softmax_temperature', c_float),
              ('initial_h', c_int), ('initial_w', c_int), ('dtype', c_int)]


class TapirNextNorm(ctypes.Structure):   # tapir_next_norm (include/tapir_hip.h)
  _fields_ = [('gamma', c_void_p), ('beta', c_void_p), ('ss', c_void_p), ('arrive', c_void_p)]


class TapirPyramid(ctypes.Structure):
  _fields_ = [('n_levels', c_int), ('query', c_void_p * 3), ('grid', c_void_p * 3),
              ('h', c_int * 3), ('w', c_int * 3), ('C', c_int * 3)]


class TapirTrajArgs(ctypes.Structure):
  _L = TAPIR_MAX_LEVELS
  _fields_ = [('B', c_int), ('Q', c_int), ('T', c_int), ('n_levels', c_int),
              ('lowres', c_void_p * _L), ('hires', c_void_p * _L),
              ('lowres_h', c_int * _L), ('lowres_w', c_int * _L),
              ('hires_h', c_int * _L), ('hires_w', c_int * _L),
              ('res_h', c_int * _L), ('res_w', c_int * _L),
              ('q_lowres', c_void_p * _L), ('q_hires', c_void_p * _L),
              ('query_points', c_void_p), ('video_h', c_int), ('video_w', c_int),
              ('ctx1_in', c_void_p), ('ctx2_in', c_void_p),
              ('ctx1_out', c_void_p), ('ctx2_out', c_void_p),
              ('tracks', c_void_p), ('occlusion', c_void_p), ('expected_dist', c_void_p)]


# name -> (restype, argtypes); every symbol include/tapir_hip.h declares
PROTOTYPES = {
    'tapir_create': (c_int, [POINTER(c_void_p), POINTER(TapirCfg), c_int]),
    'tapir_destroy': (None, [c_void_p]),
    'tapir_last_error': (c_char_p, [c_void_p]),
    'tapir_version': (c_char_p, []),
    'tapir_set_weight': (c_int, [c_void_p, c_char_p, c_void_p, POINTER(c_int64), c_int]),
    'tapir_finalize_weights': (c_int, [c_void_p]),
    'tapir_reserve': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int]),
    'tapir_pin_workspaces': (c_int, [c_void_p, c_int]),
    'tapir_build_cost_volume': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_void_p, c_void_p]),
    'tapir_tracks_from_cost_volume': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                              c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p]),
    'tapir_tapnet_tracks_from_cost_volume': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                                     c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'tapir_get_query_features': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                         c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'tapir_pips_mixer': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    'tapir_refine_pips': (c_int, [c_void_p, POINTER(TapirPyramid), c_int, c_int, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p]),
    'tapir_estimate_trajectories': (c_int, [c_void_p, POINTER(TapirTrajArgs), c_void_p]),
    'tapir_inorm_stats': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                  c_int, c_void_p]),
    'tapir_inorm_relu': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'tapir_l2_normalize': (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    'tapir_l2_normalize_staged': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_void_p]),
    'tapir_cycle_consistency_tracks': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tapir_set_staged_grid': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'tapir_clear_staged_grids': (c_int, [c_void_p]),
    'tapir_conv_plan': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'tapir_conv_set_small': (c_int, [c_void_p, c_int]),
    'tapir_conv_pack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    'tapir_conv_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_int, c_int, c_void_p]),
    'tapir_stem_plan': (c_int, [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'tapir_stem_pack': (c_int, [c_void_p, c_void_p, POINTER(c_void_p)]),
    'tapir_conv_free': (c_int, [c_void_p, c_void_p]),
    'tapir_layernorm_affine': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    'tapir_xconv_plan': (c_int, [c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'tapir_xconv_plan_frames': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int),
                                        POINTER(c_int)]),
    'tapir_xconv_pack': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    'tapir_xconv': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_void_p]),
    'tapir_xconv_nt': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_void_p]),
    'tapir_stem_conv': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tapir_stem_conv_nn': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   POINTER(TapirNextNorm), c_void_p]),
    'tapir_conv_fused_nn': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, POINTER(TapirNextNorm), c_void_p]),
    'tapir_conv_pack_dual': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_void_p)]),
    'tapir_conv_fused_dual_nn': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_int, POINTER(TapirNextNorm), c_void_p]),
    'tapir_debug_gemm': (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p,
                                 c_long, c_void_p, c_long, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p]),
    'tapir_debug_set_trace': (c_int, [c_void_p, c_void_p]),
    'tapir_debug_set_mixer_mode': (c_int, [c_void_p, c_int]),
    'tapir_debug_set_cv_mode': (c_int, [c_void_p, c_int]),
    'tapir_debug_set_patch_mode': (c_int, [c_void_p, c_int]),
    'tapir_debug_contraction': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'tapir_debug_set_gemm_mode': (c_int, [c_void_p, c_int]),
    'tapir_online_sync_error': (c_int, [c_void_p, POINTER(ctypes.c_uint)]),
    'tapir_debug_set_update_mode': (c_int, [c_void_p, c_int]),
    'tapir_debug_set_conv_flat': (c_int, [c_void_p, c_int]),
    'tapir_conv_flat_plan': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    'tapir_debug_mixer_stop': (c_int, [c_void_p, c_int]),
    'tapir_debug_workspace': (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(ctypes.c_ulonglong)]),
    'tapir_debug_poison_lds': (c_int, [c_void_p, ctypes.c_uint, c_void_p]),
    'tapir_debug_mix': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                c_void_p]),
    'tapir_profile_enable': (c_int, [c_void_p, c_int]),
    'tapir_profile_stride': (c_int, [c_void_p, c_int]),
    'tapir_profile_read': (c_int, [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(c_int64)]),
}

PROF_KINDS = {'gemm_up': 0, 'gemm_down': 1, 'mix': 2, 'patch_corr': 3, 'cv_heads': 4, 'cv_gemm': 5,
              'mixer_fused': 6, 'stem': 7, 'conv3x3_c64': 8, 'conv3x3_c128': 9, 'conv3x3_c256': 10, 'conv_other': 11,
              'l2norm': 12}


def declare_prototypes(lib: ctypes.CDLL) -> ctypes.CDLL:
  """Attaches restype/argtypes for every exported entry point; raises if one is missing."""
  for name, (restype, argtypes) in PROTOTYPES.items():
    fn = getattr(lib, name)  # AttributeError if the symbol is not exported
    fn.restype = restype
    fn.argtypes = argtypes
  return lib


_LIB = None


def load_library() -> ctypes.CDLL:
  """Loads the in-tree gfx950 library.  Fails loudly when it has not been built."""
  global _LIB
  if _LIB is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError(
          f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; '
          'g.build()"` (hipcc --offload-arch=gfx950).  tapnet_amd has no CPU fallback.')
    _LIB = declare_prototypes(ctypes.CDLL(LIB_PATH))
  return _LIB


class TapirError(RuntimeError):
  pass


def check(lib, ctx, rc: int, what: str):
  """Maps a C-ABI return code to the exception the reference would raise."""
  if rc == TAPIR_OK:
    return
  msg = lib.tapir_last_error(ctx).decode() if ctx else ''
  text = f'{what} failed ({rc}): {msg}'
  if rc in (TAPIR_ERR_INVALID, TAPIR_ERR_UNSUPPORTED):
    raise ValueError(text)
  raise TapirError(text)
