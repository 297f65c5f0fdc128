"""TEST / BASELINE INFRASTRUCTURE -- plain PyTorch restatement of the feature backbone
(TAPIR.get_feature_grids, tapnet/models/tapir_model.py:626-729): bilinear resize -> ResNet-v2 with
InstanceNorm (tapnet/models/resnet.py:150-257, 305-471; strides (1,2,2,1), channels
(64,128,256,256), 2 blocks per group, projection shortcut on the first block of each group, 7x7/2
stem) -> optional ExtraConvs (tapir_model.py:159-186) on the low-res map -> per-pixel L2
normalisation (:709-720).  Convolutions use XLA "SAME" padding (extra pad on the high side for stride
2).  Weights are looked up by the reference's torch state_dict names (tapnet/torch/nets.py).

Runs on the CPU.  Used by tests/test_host_logic.py (held to the reference's own feature grids in
tests/golden/backbone.npz) and by bench.py's cpu_baseline leg -- never by the product, whose
backbone (tapnet_amd/backbone.py) is GPU-only: MIOpen convolutions + HIP kernels.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


def _same_pad(x: torch.Tensor, k: int, stride: int) -> torch.Tensor:
  """XLA SAME padding for an NCHW tensor: total = max((ceil(n/s)-1)*s+k-n, 0), low = total//2."""
  h, w = x.shape[-2:]
  th = max((-(-h // stride) - 1) * stride + k - h, 0)
  tw = max((-(-w // stride) - 1) * stride + k - w, 0)
  if th == 0 and tw == 0:
    return x
  return F.pad(x, (tw // 2, tw - tw // 2, th // 2, th - th // 2))


class TorchBackbone:
  """Backbone weights on the CPU in float32."""

  def __init__(self, weights: Dict[str, torch.Tensor], extra_convs: bool,
               blocks_per_group: Sequence[int] = (2, 2, 2, 2)):
    self.extra_convs = extra_convs
    self.blocks_per_group = tuple(blocks_per_group)
    self.w: Dict[str, torch.Tensor] = {}
    for k, v in weights.items():
      if k.startswith('resnet_torch.') or k.startswith('extra_convs.'):
        self.w[k] = torch.as_tensor(v).float()

  # -- building blocks ------------------------------------------------------
  def _conv(self, x, name, stride=1, bias=False):
    w = self.w[name + '.weight']
    b = self.w[name + '.bias'].to(x.dtype) if bias else None
    return F.conv2d(_same_pad(x, w.shape[-1], stride), w, b, stride=stride)

  def _inorm_relu(self, x, name):
    # hk.InstanceNorm(create_scale, create_offset), eps 1e-5 (resnet.py:177-181); stats in f32
    y = F.instance_norm(x.float(), weight=self.w[name + '.weight'], bias=self.w[name + '.bias'],
                        eps=1e-5)
    return torch.relu(y).to(x.dtype)

  def _block(self, x, p, stride, use_projection):
    shortcut = x
    y = self._inorm_relu(x, p + 'bn_0')
    if use_projection:
      shortcut = self._conv(y, p + 'proj_conv', stride)
    y = self._conv(y, p + 'conv_0', stride)
    y = self._inorm_relu(y, p + 'bn_1')
    y = self._conv(y, p + 'conv_1', 1)
    return y + shortcut

  def _resnet(self, x):
    x = self._conv(x, 'resnet_torch.initial_conv', 2)
    outs = {}
    strides = (1, 2, 2, 1)
    for g in range(4):
      for b in range(self.blocks_per_group[g]):
        x = self._block(x, f'resnet_torch.block_groups.{g}.blocks.{b}.',
                        strides[g] if b == 0 else 1, b == 0)
      outs[g] = x
    return outs[3], outs[1]   # resnet_unit_3 (256 ch, /8), resnet_unit_1 (128 ch, /4)

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

  @staticmethod
  def _l2norm(x_nhwc):
    s = torch.sum(torch.square(x_nhwc), dim=-1, keepdim=True)
    return x_nhwc / torch.sqrt(torch.clamp_min(s, 1e-12))

  @torch.no_grad()
  def features(self, frames_nhwc: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """frames [N,H,W,3] f32 in [-1,1] -> (lowres [N,H/8,W/8,256], hires [N,H/4,W/4,128]) f32,
    L2-normalised, channels-last."""
    x = torch.as_tensor(frames_nhwc).float().permute(0, 3, 1, 2)
    u3, u1 = self._resnet(x)
    if self.extra_convs:
      u3 = self._extra_convs(u3)
    return (self._l2norm(u3.permute(0, 2, 3, 1)).contiguous(),
            self._l2norm(u1.permute(0, 2, 3, 1)).contiguous())
