"""Host-side helpers mirroring tapnet/utils/model_utils.py:317-389 and
tapnet/utils/transforms.py:24-78 (pure plumbing; no array compute on the hot path)."""
from __future__ import annotations

from typing import Sequence

import numpy as np


def convert_grid_coordinates(coords, input_grid_size: Sequence[int],
                             output_grid_size: Sequence[int], coordinate_format: str = 'xy'):
  """coords * out / in (transforms.py:75-76; the +-0.5 of its docstring is not implemented
  there either).  Works on numpy arrays and torch tensors."""
  input_grid_size = np.asarray(input_grid_size)
  output_grid_size = np.asarray(output_grid_size)
  if coordinate_format == 'xy':
    if input_grid_size.shape[0] != 2 or output_grid_size.shape[0] != 2:
      raise ValueError('If coordinate_format is xy, the shapes must be length 2.')
  elif coordinate_format == 'tyx':
    if input_grid_size.shape[0] != 3 or output_grid_size.shape[0] != 3:
      raise ValueError('If coordinate_format is tyx, the shapes must be length 3.')
    if input_grid_size[0] != output_grid_size[0]:
      raise ValueError('converting frame count is not supported.')
  else:
    raise ValueError('Recognized coordinate formats are xy and tyx.')
  scale = (output_grid_size / input_grid_size).astype(np.float32)
  if isinstance(coords, np.ndarray):
    return (coords * output_grid_size.astype(coords.dtype)
            / input_grid_size.astype(coords.dtype)).astype(coords.dtype)
  import torch
  return coords * torch.as_tensor(scale, device=coords.device, dtype=coords.dtype)


def is_same_res(r1, r2) -> bool:
  return all(x == y for x, y in zip(r1, r2))


def generate_default_resolutions(full_size, train_size, num_levels=None):
  """model_utils.py:317-359."""
  if all(x == y for x, y in zip(train_size, full_size)):
    return [tuple(train_size)]
  if num_levels is None:
    size_ratio = np.array(full_size) / np.array(train_size)
    num_levels = int(np.ceil(np.max(np.log2(size_ratio))) + 1)
  if num_levels <= 1:
    return [tuple(train_size)]
  h, w = full_size[0:2]
  if h % 8 != 0 or w % 8 != 0:
    print('Warning: output size is not a multiple of 8. Final layer will round size down.')
  ll_h, ll_w = train_size[0:2]
  sizes = []
  for i in range(num_levels):
    sizes.append((
        int(round((ll_h * (h / ll_h) ** (i / (num_levels - 1))) // 8)) * 8,
        int(round((ll_w * (w / ll_w) ** (i / (num_levels - 1))) // 8)) * 8,
    ))
  return sizes


def preprocess_frames(frames):
  """uint8 [0,255] -> float32 [-1,1] (model_utils.py:362-373)."""
  frames = np.asarray(frames).astype(np.float32)
  return frames / 255 * 2 - 1


def postprocess_occlusions(occlusions, expected_dist):
  """visible iff (1-sigmoid(occ))(1-sigmoid(expd)) > 0.5 (model_utils.py:376-389)."""
  occlusions, expected_dist = np.asarray(occlusions), np.asarray(expected_dist)
  sig = lambda v: 1.0 / (1.0 + np.exp(-v))
  return (1 - sig(occlusions)) * (1 - sig(expected_dist)) > 0.5
