"""Checkpoint plumbing: flat torch-named weight dicts (the reference's ``.pt``
state_dict, tapnet/torch/tapir_model.py:115-137) and Haiku ``.npy`` params
(``np.load(path, allow_pickle=True).item()['params']``, live_demo.py:31-33).

The Haiku -> torch-name conversion follows the module names in
tapnet/models/tapir_model.py:341-384 and tapnet/models/resnet.py:185-222,
399-448.  It is held to the parameter tree the reference's own modules create
when run over numpy stand-ins for jax / haiku (oracle/make_jax_golden.py,
tests/test_jax_reference_pin.py); a real released ``.npy`` could not be fetched
offline.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping

import numpy as np


def is_haiku_params(params) -> bool:
  if not isinstance(params, Mapping) or not params:
    return False
  k = next(iter(params))
  return isinstance(params[k], Mapping)


def load_checkpoint(path: str) -> Dict[str, np.ndarray]:
  """Loads a reference checkpoint (.pt state_dict or Haiku .npy) as a torch-named flat dict."""
  if path.endswith('.npy'):
    ckpt = np.load(path, allow_pickle=True).item()
    return haiku_to_torch_names(ckpt['params'] if 'params' in ckpt else ckpt)
  import torch
  sd = torch.load(path, map_location='cpu')
  return {k: v.numpy() for k, v in sd.items()}


def to_torch_names(params) -> Dict[str, Any]:
  return haiku_to_torch_names(params) if is_haiku_params(params) else dict(params)


def _conv_w(w):  # Haiku HWIO -> torch OIHW
  return np.ascontiguousarray(np.transpose(np.asarray(w), (3, 2, 0, 1)))


def haiku_to_torch_names(params: Mapping[str, Mapping[str, Any]]) -> Dict[str, np.ndarray]:
  out: Dict[str, np.ndarray] = {}

  def get(mod, name):
    return np.asarray(params[mod][name], dtype=np.float32)

  root = 'tapir/~/'
  # cost-volume heads (tapir_model.py:341-361)
  cv = {'hid1': 'cost_volume_regression_1', 'hid2': 'cost_volume_regression_2',
        'hid3': 'cost_volume_occlusion_1'}
  for t, h in cv.items():
    out[f'torch_cost_volume_track_mods.{t}.weight'] = _conv_w(get(root + h, 'w'))
    out[f'torch_cost_volume_track_mods.{t}.bias'] = get(root + h, 'b')
  for t, h in {'hid4': 'cost_volume_occlusion_2', 'occ_out': 'occlusion_out'}.items():
    out[f'torch_cost_volume_track_mods.{t}.weight'] = get(root + h, 'w').T.copy()
    out[f'torch_cost_volume_track_mods.{t}.bias'] = get(root + h, 'b')
  # mixer (tapir_model.py:127-156, 39-98)
  mx = root + 'pips_mlp_mixer/'
  out['torch_pips_mixer.linear.weight'] = get(mx + 'linear', 'w').T.copy()
  out['torch_pips_mixer.linear.bias'] = get(mx + 'linear', 'b')
  out['torch_pips_mixer.linear_1.weight'] = get(mx + 'linear_1', 'w').T.copy()
  out['torch_pips_mixer.linear_1.bias'] = get(mx + 'linear_1', 'b')
  out['torch_pips_mixer.layer_norm.weight'] = get(mx + 'layer_norm', 'scale')
  i = 0
  while True:
    blk = mx + ('block' if i == 0 else f'block_{i}') + '/'
    if blk + 'mlp1_up' not in params:
      break
    p = f'torch_pips_mixer.blocks.{i}.'
    out[p + 'layer_norm.weight'] = get(blk + 'layer_norm', 'scale')
    out[p + 'layer_norm_1.weight'] = get(blk + 'layer_norm_1', 'scale')
    for t, h in (('mlp1_up', 'mlp1_up'), ('mlp1_up_1', 'mlp1_up_1')):
      w = get(blk + h, 'w')            # hk.DepthwiseConv1D: [k, 1, C*mult]
      out[p + t + '.weight'] = np.ascontiguousarray(np.transpose(w, (2, 1, 0)))
      out[p + t + '.bias'] = get(blk + h, 'b').reshape(-1)
    for t in ('mlp2_up', 'mlp2_down'):
      out[p + f'conv_channels_mixer.{t}.weight'] = get(blk + t, 'w').T.copy()
      out[p + f'conv_channels_mixer.{t}.bias'] = get(blk + t, 'b')
    i += 1
  # backbone (resnet.py:399-448, 185-222)
  rn = root + 'resnet/~/'
  if rn + 'initial_conv' in params:
    out['resnet_torch.initial_conv.weight'] = _conv_w(get(rn + 'initial_conv', 'w'))
    for g in range(4):
      b = 0
      while True:
        blk = f'{rn}block_group_{g}/~/block_{b}/~/'
        if blk + 'conv_0' not in params:
          break
        p = f'resnet_torch.block_groups.{g}.blocks.{b}.'
        if blk + 'shortcut_conv' in params:
          out[p + 'proj_conv.weight'] = _conv_w(get(blk + 'shortcut_conv', 'w'))
        for j in (0, 1):
          out[p + f'conv_{j}.weight'] = _conv_w(get(blk + f'conv_{j}', 'w'))
          out[p + f'bn_{j}.weight'] = get(blk + f'instancenorm_{j}', 'scale').reshape(-1)
          out[p + f'bn_{j}.bias'] = get(blk + f'instancenorm_{j}', 'offset').reshape(-1)
        b += 1
  ec = root + 'extra_convs/'
  n = 0
  while True:
    ln = ec + ('layer_norm' if n == 0 else f'layer_norm_{n}')
    if ln not in params:
      break
    p = f'extra_convs.blocks.{n}.'
    out[p + 'layer_norm.weight'] = get(ln, 'scale')
    out[p + 'layer_norm.bias'] = get(ln, 'offset')
    c0 = ec + ('conv2_d' if n == 0 else f'conv2_d_{2 * n}')
    c1 = ec + f'conv2_d_{2 * n + 1}'
    out[p + 'conv.weight'] = _conv_w(get(c0, 'w')); out[p + 'conv.bias'] = get(c0, 'b')
    out[p + 'conv_1.weight'] = _conv_w(get(c1, 'w')); out[p + 'conv_1.bias'] = get(c1, 'b')
    n += 1
  return out


def torch_to_haiku_names(weights: Mapping[str, Any]) -> Dict[str, Dict[str, np.ndarray]]:
  """Inverse of haiku_to_torch_names: a flat torch-named dict -> the Haiku tree ``{module: {leaf: array}}``
  the reference's ``ParameterizedTAPIR(params, state, tapir_kwargs)`` consumes (tapir_model.py:1199-1260).
  Used to run the reference's JAX text on this package's weights (oracle/make_jax_golden.py) and to export."""
  w = {k: np.asarray(v, dtype=np.float32) for k, v in weights.items()}
  out: Dict[str, Dict[str, np.ndarray]] = {}
  hwio = lambda a: np.ascontiguousarray(np.transpose(a, (2, 3, 1, 0)))
  root = 'tapir/~/'
  for t, h in (('hid1', 'cost_volume_regression_1'), ('hid2', 'cost_volume_regression_2'),
               ('hid3', 'cost_volume_occlusion_1')):
    out[root + h] = {'w': hwio(w[f'torch_cost_volume_track_mods.{t}.weight']),
                     'b': w[f'torch_cost_volume_track_mods.{t}.bias']}
  for t, h in (('hid4', 'cost_volume_occlusion_2'), ('occ_out', 'occlusion_out')):
    out[root + h] = {'w': w[f'torch_cost_volume_track_mods.{t}.weight'].T.copy(),
                     'b': w[f'torch_cost_volume_track_mods.{t}.bias']}
  mx = root + 'pips_mlp_mixer/'
  for t in ('linear', 'linear_1'):
    out[mx + t] = {'w': w[f'torch_pips_mixer.{t}.weight'].T.copy(), 'b': w[f'torch_pips_mixer.{t}.bias']}
  out[mx + 'layer_norm'] = {'scale': w['torch_pips_mixer.layer_norm.weight']}
  i = 0
  while f'torch_pips_mixer.blocks.{i}.mlp1_up.weight' in w:
    p = f'torch_pips_mixer.blocks.{i}.'
    blk = mx + ('block' if i == 0 else f'block_{i}') + '/'
    out[blk + 'layer_norm'] = {'scale': w[p + 'layer_norm.weight']}
    out[blk + 'layer_norm_1'] = {'scale': w[p + 'layer_norm_1.weight']}
    for t in ('mlp1_up', 'mlp1_up_1'):   # torch depthwise Conv1d [C*mult, 1, k] -> hk.DepthwiseConv1D [k, 1, C*mult]
      out[blk + t] = {'w': np.ascontiguousarray(np.transpose(w[p + t + '.weight'], (2, 1, 0))),
                      'b': w[p + t + '.bias']}
    for t in ('mlp2_up', 'mlp2_down'):
      out[blk + t] = {'w': w[p + f'conv_channels_mixer.{t}.weight'].T.copy(),
                      'b': w[p + f'conv_channels_mixer.{t}.bias']}
    i += 1
  rn = root + 'resnet/~/'
  if 'resnet_torch.initial_conv.weight' in w:
    out[rn + 'initial_conv'] = {'w': hwio(w['resnet_torch.initial_conv.weight'])}
    g = 0
    while f'resnet_torch.block_groups.{g}.blocks.0.conv_0.weight' in w:
      b = 0
      while f'resnet_torch.block_groups.{g}.blocks.{b}.conv_0.weight' in w:
        p = f'resnet_torch.block_groups.{g}.blocks.{b}.'
        blk = f'{rn}block_group_{g}/~/block_{b}/~/'
        if p + 'proj_conv.weight' in w:
          out[blk + 'shortcut_conv'] = {'w': hwio(w[p + 'proj_conv.weight'])}
        for j in (0, 1):
          out[blk + f'conv_{j}'] = {'w': hwio(w[p + f'conv_{j}.weight'])}
          out[blk + f'instancenorm_{j}'] = {'scale': w[p + f'bn_{j}.weight'], 'offset': w[p + f'bn_{j}.bias']}
        b += 1
      g += 1
  n = 0
  while f'extra_convs.blocks.{n}.conv.weight' in w:
    p = f'extra_convs.blocks.{n}.'
    ec = root + 'extra_convs/'
    out[ec + ('layer_norm' if n == 0 else f'layer_norm_{n}')] = {'scale': w[p + 'layer_norm.weight'],
                                                                'offset': w[p + 'layer_norm.bias']}
    out[ec + ('conv2_d' if n == 0 else f'conv2_d_{2 * n}')] = {'w': hwio(w[p + 'conv.weight']), 'b': w[p + 'conv.bias']}
    out[ec + f'conv2_d_{2 * n + 1}'] = {'w': hwio(w[p + 'conv_1.weight']), 'b': w[p + 'conv_1.bias']}
    n += 1
  return out
