#!/usr/bin/env python
"""CPU experiment (no GPU needed): WHICH of the bf16 backbone's roundings cost the accuracy?

Emulates the rounding points of the HIP bf16 backbone (csrc/conv_fused.hpp: operands of every convolution --
normalised activations and weights -- rounded to bf16, f32 accumulation, stored tensors rounded to bf16) in
float32 torch on the CPU, and toggles them:
  A  what the bf16 build does today (residual stream, block-internal tensor and operands in bf16)
  B  residual stream (stem output, block outputs, projection outputs) kept in f32; operands still bf16
  C  every stored tensor f32, only the MFMA operands bf16 (the floor of a bf16-MFMA backbone)
against the f32 backbone: per-pixel cosine of the low-res grid and the argmax flip rate of the cost volume
(f32 arithmetic downstream) on the moving-texture clip.  VERDICT r2 item 4(a)."""
import json
import sys
import os

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import tapir_oracle as O  # noqa: E402
from oracle.backbone_torch import _same_pad  # noqa: E402
from tapnet_amd import synthetic  # noqa: E402

bf = lambda t: t.to(torch.bfloat16).float()
ident = lambda t: t


def backbone(w, frames, r_op, r_res, r_mid):
  """r_op: rounding of MFMA operands; r_res: of the residual stream; r_mid: of conv_0's stored output."""
  W = {k: torch.as_tensor(v).float() for k, v in w.items() if k.startswith('resnet_torch.')}
  conv = lambda x, name, s: F.conv2d(_same_pad(x, W[name + '.weight'].shape[-1], s), r_op(W[name + '.weight']), None, stride=s)

  def nrm(x, name):   # statistics of the STORED tensor; the operand is rounded once, after norm + relu
    return r_op(torch.relu(F.instance_norm(x, weight=W[name + '.weight'], bias=W[name + '.bias'], eps=1e-5)))

  x = r_res(conv(r_op(frames.permute(0, 3, 1, 2)), 'resnet_torch.initial_conv', 2))
  outs = {}
  for g, st in enumerate((1, 2, 2, 1)):
    for b in range(2):
      p = f'resnet_torch.block_groups.{g}.blocks.{b}.'
      s = st if b == 0 else 1
      y = nrm(x, p + 'bn_0')
      sc = r_res(conv(y, p + 'proj_conv', s)) if b == 0 else x
      y0 = r_mid(conv(y, p + 'conv_0', s))
      x = r_res(conv(nrm(y0, p + 'bn_1'), p + 'conv_1', 1) + sc)
    outs[g] = x
  l2 = lambda t: (t / torch.sqrt(torch.clamp_min((t * t).sum(1, keepdim=True), 1e-12))).permute(0, 2, 3, 1).contiguous()
  return l2(outs[3]).numpy(), l2(outs[1]).numpy()


def main():
  T, S, Q = 8, 256, 192
  torch.set_num_threads(os.cpu_count() or 1)
  w = synthetic.make_weights(3, 0, False)
  video = synthetic.make_video(7, T, S, S)
  qp = synthetic.make_queries(8 + 256, Q, T, S, S)
  fr = torch.as_tensor(video[0])
  ref_low, ref_hi = backbone(w, fr, ident, ident, ident)

  def cv_argmax(low):
    ql, _ = O.get_query_features([low[None]], [low[None][..., :128]], [(S, S)], qp, video.shape)
    _, _, _, st = O.tracks_from_cost_volume(w, ql[0], low[None], qp, (S, S), 20.0, return_stages=True)
    return st['argmax'], st['top2_rel_gap']

  a_ref, gap = cv_argmax(ref_low)
  out = {}
  for name, (rop, rres, rmid) in dict(A_all_bf16=(bf, bf, bf), B_f32_residual=(bf, ident, bf),
                                      C_operands_only=(bf, ident, ident)).items():
    low, hi = backbone(w, fr, rop, rres, rmid)
    cos = (low * ref_low).sum(-1)
    a, _ = cv_argmax(low)
    out[name] = dict(lowres_cos_min=float(cos.min()), lowres_cos_median=float(np.median(cos)),
                     hires_cos_min=float((hi * ref_hi).sum(-1).min()),
                     argmax_flip_rate=float((a != a_ref).mean()),
                     argmax_flip_rate_clear=float((a != a_ref)[gap > 1e-2].mean()))
    print(name, out[name], flush=True)
  out['what'] = f'{S}x{S}x{T} moving-texture clip, {Q} queries, random-init TAPIR weights; flips of the cost-volume argmax vs the f32 backbone'
  print(json.dumps(out))


if __name__ == '__main__':
  main()
