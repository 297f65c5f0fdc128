"""CPU experiment (numpy oracle, no GPU): what would fp8 weights in the mixer's channel MLP cost in track accuracy?
The mixer is bound by the bytes of W_up / W_down streamed per track; fp8 (e4m3, one f32 scale per output
row, converted to bf16 in registers) would halve them.  This measures the deviation it adds, next to the deviation the
bf16 build already has against f32 -- hot path only, synthetic grids, random-init weights (peaky heat maps).

    python tools/exp_fp8_weights.py > profiles/r03_fp8_weight_experiment.json
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tapir_oracle as O   # noqa: E402
from tapnet_amd import synthetic       # noqa: E402


def fp8_e4m3(x):
  """Round to the nearest e4m3 value (4 exponent bits, bias 7, 3 mantissa bits, max 448, subnormals down to 2^-9)."""
  x = np.asarray(x, np.float64)
  s, a = np.sign(x), np.minimum(np.abs(x), 448.0)
  e = np.floor(np.log2(np.maximum(a, 2.0 ** -9)))
  e = np.maximum(e, -6.0)                       # subnormal range shares the exponent of 2^-6
  q = 2.0 ** (e - 3)
  return (s * np.round(a / q) * q).astype(np.float32)


def quantise_rows(w):
  scale = np.abs(w).max(axis=1, keepdims=True) / 448.0
  return (fp8_e4m3(w / scale) * scale).astype(np.float32)


def quantise_matrix(w):
  """ONE scale per matrix: it folds into the LayerNorm scale (W_up) / the GELU output (W_down) -- no run-time cost."""
  scale = np.abs(w).max() / 448.0
  return (fp8_e4m3(w / scale) * scale).astype(np.float32)


def run(w, lows, his, res, qp, shape, rnd):
  return O.tapir_from_grids(w, shape, lows, his, res, qp, pyramid_level=0, softmax_temperature=20.0,
                            initial_resolution=res[0], rnd=rnd)


def main():
  rng = np.random.default_rng(0)
  T, Q, S = 24, 48, 64
  w = synthetic.make_weights(3, 0, False, backbone=False)
  low = O.l2_normalize(rng.standard_normal((1, T, S // 8, S // 8, 256)).astype(np.float32))
  hi = O.l2_normalize(rng.standard_normal((1, T, S // 4, S // 4, 128)).astype(np.float32))
  qp = synthetic.make_queries(5, Q, T, S, S)
  shape, res = (1, T, S, S, 3), [(S, S), (S, S)]
  f32 = run(w, [low, low], [hi, hi], res, qp, shape, None)
  b16 = run(w, [low, low], [hi, hi], res, qp, shape, O.bf16_round)
  w8 = dict(w)
  nq = 0
  for k in w:
    if 'conv_channels_mixer.mlp2_' in k and k.endswith('.weight'):
      w8[k] = quantise_rows(w[k]); nq += 1
  f8 = run(w8, [low, low], [hi, hi], res, qp, shape, O.bf16_round)
  w8m = dict(w)
  for k in w:
    if 'conv_channels_mixer.mlp2_' in k and k.endswith('.weight'):
      w8m[k] = quantise_matrix(w[k])
  f8m = run(w8m, [low, low], [hi, hi], res, qp, shape, O.bf16_round)
  dev = lambda a, b: np.linalg.norm(np.asarray(a['tracks']) - np.asarray(b['tracks']), axis=-1).ravel()
  stat = lambda d: dict(median_px=float(np.median(d)), p99_px=float(np.percentile(d, 99)), max_px=float(d.max()))
  occ = lambda a, b: float(np.abs(np.asarray(a['occlusion']) - np.asarray(b['occlusion'])).max())
  print(json.dumps(dict(
      workload=f'hot path, {T} frames, {Q} queries, {S}x{S}, 4 iterations, random-init weights; oracle with the bf16 '
               'build\'s operand roundings', quantised_matrices=nq,
      bf16_vs_f32=dict(**stat(dev(b16, f32)), occlusion_logit_max=occ(b16, f32)),
      fp8_weights_vs_bf16=dict(**stat(dev(f8, b16)), occlusion_logit_max=occ(f8, b16)),
      fp8_weights_vs_f32=dict(**stat(dev(f8, f32)), occlusion_logit_max=occ(f8, f32)),
      fp8_one_scale_per_matrix_vs_bf16=dict(**stat(dev(f8m, b16)), occlusion_logit_max=occ(f8m, b16)),
      fp8_one_scale_per_matrix_vs_f32=dict(**stat(dev(f8m, f32)), occlusion_logit_max=occ(f8m, f32))), indent=1))


if __name__ == '__main__':
  main()
