"""TEST INFRASTRUCTURE ONLY -- numpy restatement of ``jax.image.resize(x, shape, method='bilinear')`` as the JAX
model calls it in TAPIR.get_feature_grids (tapnet/models/tapir_model.py:670, via tapnet/utils/model_utils /
`jax.image.resize(video, ..., 'bilinear')`).

Parity status: UNPINNED against an execution of JAX (jax / jaxlib are not installable offline; the reference pins no
version: pyproject.toml:5-19 lists `jax` without a bound).  What is restated is the PUBLISHED algorithm of
`jax/_src/image/scale.py` (`_resize` -> `_scale_and_translate` -> `compute_weight_mat`, unchanged in substance since
jax 0.2.x):

  * per spatial axis a dense weight matrix W [in, out] is built and applied as a matrix product (separable);
  * with `antialias=True` (the default of jax.image.resize) the triangle kernel is widened by the down-scaling factor:
        inv_scale = in / out;  kernel_scale = max(inv_scale, 1)
        sample_f  = (arange(out) + 0.5) * inv_scale - 0.5          (centre of output pixel j in input coordinates)
        x         = |sample_f[None, :] - arange(in)[:, None]| / kernel_scale
        w         = max(0, 1 - x)                                   (triangle kernel, radius 1)
        W         = w / sum_in(w)   (0 where the sum is below 1000 eps)
        W[:, j]   = 0 where sample_f[j] is outside [-0.5, in - 0.5]
    so up-sampling (kernel_scale = 1) is plain bilinear interpolation with edge renormalisation -- equal to
    torch's F.interpolate(bilinear, align_corners=False) -- and down-sampling averages over a support of 2 * inv_scale
    input pixels, which the reference's torch twin (tapnet/torch/utils.py:39, no antialias) does not.

The product implements the JAX behaviour with F.interpolate(..., antialias=True) (tapnet_amd/backbone.py
resize_bilinear, opt-in `jax_antialias_resize=True`); tests/test_host_logic.py holds it to this restatement."""
import numpy as np


def compute_weight_mat(in_size: int, out_size: int, antialias: bool = True) -> np.ndarray:
  inv_scale = in_size / out_size
  kernel_scale = max(inv_scale, 1.0) if antialias else 1.0
  sample_f = (np.arange(out_size, dtype=np.float64) + 0.5) * inv_scale - 0.5
  x = np.abs(sample_f[None, :] - np.arange(in_size, dtype=np.float64)[:, None]) / kernel_scale
  w = np.maximum(0.0, 1.0 - x)
  total = w.sum(axis=0, keepdims=True)
  w = np.where(np.abs(total) > 1000.0 * np.finfo(np.float32).eps, w / np.where(total != 0, total, 1), 0.0)
  inside = (sample_f >= -0.5) & (sample_f <= in_size - 0.5)
  return np.where(inside[None, :], w, 0.0)


def resize_bilinear(video: np.ndarray, resolution, antialias: bool = True) -> np.ndarray:
  """[B, T, H, W, C] -> [B, T, h, w, C] like jax.image.resize(video, (B, T, h, w, C), 'bilinear')."""
  h, w = int(resolution[0]), int(resolution[1])
  wy = compute_weight_mat(video.shape[2], h, antialias)
  wx = compute_weight_mat(video.shape[3], w, antialias)
  out = np.einsum('bthwc,hi->btiwc', video.astype(np.float64), wy)
  out = np.einsum('btiwc,wj->btijc', out, wx)
  return out.astype(np.float32)
