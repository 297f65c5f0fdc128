"""-m gpu: the fused 3x3 backbone convolution (csrc/conv_fused.hpp) through the C ABI on the real
gfx950 library against a plain PyTorch f32 reference of the same op (resnet.py:241-256: InstanceNorm,
relu, 3x3 SAME convolution, residual add), and the backbone with / without it."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tapnet_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def model():
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(21, 1, False)
  return tapir_model.TAPIR(pyramid_level=1, extra_convs=False, weights=w, device='cuda:0', dtype='bfloat16')


@pytest.mark.parametrize('n,h,w,c,shortcut', [(4, 128, 128, 64, True), (3, 64, 64, 128, True), (5, 32, 32, 256, False),
                                              (2, 30, 40, 128, True), (2, 17, 24, 256, True), (1, 256, 256, 64, False)])
def test_conv3x3_fused_vs_torch(model, n, h, w, c, shortcut):
  lib, ctx = model._lib, model._ctx
  dev = model.device
  stream = model._stream()
  g = torch.Generator(device='cpu').manual_seed(n * 1000 + h + c)
  x = (torch.randn(n, h, w, c, generator=g) * 1.5 + 0.5).to(torch.bfloat16).to(dev)
  sc = torch.randn(n, h, w, c, generator=g).to(torch.bfloat16).to(dev) if shortcut else None
  wt = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).contiguous()
  gamma = (torch.rand(c, generator=g) + 0.5).to(dev)
  beta = (torch.randn(c, generator=g) * 0.3).to(dev)
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, h, w, c, c, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  slabs = 5
  part_in = torch.empty(n, slabs, c, 2, device=dev)
  assert lib.tapir_inorm_stats(ctx, x.data_ptr(), None, None, part_in.data_ptr(), n, h * w, c, slabs, stream) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wt.data_ptr()), c, c, 3, ctypes.byref(ws)) == 0
  y = torch.zeros(n, h, w, c, device=dev, dtype=torch.bfloat16)
  part = torch.zeros(n, tiles.value, c, 2, device=dev)
  ss = torch.empty(n, c, 2, device=dev)
  rc = lib.tapir_conv_fused(ctx, x.data_ptr(), part_in.data_ptr(), slabs, 0, gamma.data_ptr(), beta.data_ptr(),
                            ss.data_ptr(), ws, sc.data_ptr() if shortcut else None, y.data_ptr(), part.data_ptr(),
                            n, h, w, c, c, 3, 1, stream)
  assert rc == 0, lib.tapir_last_error(ctx)
  torch.cuda.synchronize()
  # f32 reference on the same bf16-rounded operands (normalised activations rounded to bf16 like the kernel's LDS image)
  xf = x.float()
  mean = xf.mean((1, 2), keepdim=True)
  var = xf.var((1, 2), keepdim=True, unbiased=False)
  xn = torch.relu((xf - mean) / torch.sqrt(var + 1e-5) * gamma + beta).to(torch.bfloat16).float()
  ref = F.conv2d(xn.permute(0, 3, 1, 2), wt.to(torch.bfloat16).float().to(dev), padding=1).permute(0, 2, 3, 1)
  if shortcut:
    ref = ref + sc.float()
  got = y.float()
  d = (got - ref).abs()
  # bf16 output rounding: half an ulp of values up to ~8 -> 1.6e-2; a normalised operand that rounds the other way
  # (the (a, b) pair comes from merged f32 summaries) adds one more operand ulp
  assert float(d.max()) < 4e-2, float(d.max())
  assert float(d.mean()) < 2.5e-3, float(d.mean())
  # the summaries describe the stored tensor
  cnt = torch.tensor([min(rows.value, h - t * rows.value) * w for t in range(tiles.value)], device=dev,
                     dtype=torch.float64)
  pm, pM2 = part[..., 0].double(), part[..., 1].double()
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  gd = got.double()
  assert torch.allclose(tot_mean, gd.mean((1, 2)), atol=1e-5)
  assert torch.allclose(tot_M2 / (h * w), gd.var((1, 2), unbiased=False), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('n,h,w,cin,cout,ks,stride', [(3, 128, 128, 64, 128, 3, 2), (3, 64, 64, 128, 256, 3, 2),
                                                      (3, 128, 128, 64, 128, 1, 2), (3, 64, 64, 128, 256, 1, 2),
                                                      (3, 128, 128, 64, 64, 1, 1), (3, 32, 32, 256, 256, 1, 1),
                                                      (2, 25, 31, 64, 128, 3, 2)])
def test_strided_and_projection_convs_vs_torch(model, n, h, w, cin, cout, ks, stride):
  lib, ctx = model._lib, model._ctx
  dev = model.device
  stream = model._stream()
  g = torch.Generator(device='cpu').manual_seed(h + cin + ks)
  x = (torch.randn(n, h, w, cin, generator=g) * 1.5 + 0.5).to(torch.bfloat16).to(dev)
  wt = (torch.randn(cout, cin, ks, ks, generator=g) / (ks * ks * cin) ** 0.5).contiguous()
  gamma = (torch.rand(cin, generator=g) + 0.5).to(dev)
  beta = (torch.randn(cin, generator=g) * 0.3).to(dev)
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, h, w, cin, cout, ks, stride, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  part_in = torch.empty(n, 4, cin, 2, device=dev)
  assert lib.tapir_inorm_stats(ctx, x.data_ptr(), None, None, part_in.data_ptr(), n, h * w, cin, 4, stream) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wt.data_ptr()), cout, cin, ks, ctypes.byref(ws)) == 0
  ho, wo = -(-h // stride), -(-w // stride)
  y = torch.zeros(n, ho, wo, cout, device=dev, dtype=torch.bfloat16)
  part = torch.zeros(n, tiles.value, cout, 2, device=dev)
  ss = torch.empty(n, cin, 2, device=dev)
  rc = lib.tapir_conv_fused(ctx, x.data_ptr(), part_in.data_ptr(), 4, 0, gamma.data_ptr(), beta.data_ptr(),
                            ss.data_ptr(), ws, None, y.data_ptr(), part.data_ptr(), n, h, w, cin, cout, ks, stride, stream)
  assert rc == 0, lib.tapir_last_error(ctx)
  torch.cuda.synchronize()
  xf = x.float()
  mean = xf.mean((1, 2), keepdim=True)
  var = xf.var((1, 2), keepdim=True, unbiased=False)
  xn = torch.relu((xf - mean) / torch.sqrt(var + 1e-5) * gamma + beta).to(torch.bfloat16).float()
  from tapnet_amd.backbone import _same_pad
  ref = F.conv2d(_same_pad(xn.permute(0, 3, 1, 2), ks, stride), wt.to(torch.bfloat16).float().to(dev),
                 stride=stride).permute(0, 2, 3, 1)
  d = (y.float() - ref).abs()
  assert float(d.max()) < 4e-2 and float(d.mean()) < 2.5e-3, (float(d.max()), float(d.mean()))
  cnt = torch.tensor([min(rows.value, ho - t * rows.value) * wo for t in range(tiles.value)], device=dev,
                     dtype=torch.float64)
  pm, pM2 = part[..., 0].double(), part[..., 1].double()
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  gd = y.double()
  assert torch.allclose(tot_mean, gd.mean((1, 2)), atol=1e-5)
  assert torch.allclose(tot_M2 / (ho * wo), gd.var((1, 2), unbiased=False), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('n,h,w', [(4, 256, 256), (2, 200, 136), (1, 512, 512), (3, 33, 50)])
def test_stem_conv_vs_torch(model, n, h, w):
  lib, ctx = model._lib, model._ctx
  dev = model.device
  g = torch.Generator(device='cpu').manual_seed(h + w)
  x = (torch.rand(n, h, w, 3, generator=g) * 2 - 1).to(dev)
  wt = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).contiguous()
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_stem_plan(ctx, h, w, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_stem_pack(ctx, ctypes.c_void_p(wt.data_ptr()), ctypes.byref(ws)) == 0
  ho, wo = -(-h // 2), -(-w // 2)
  y = torch.zeros(n, ho, wo, 64, device=dev, dtype=torch.bfloat16)
  part = torch.zeros(n, tiles.value, 64, 2, device=dev)
  rc = lib.tapir_stem_conv(ctx, x.data_ptr(), ws, y.data_ptr(), part.data_ptr(), n, h, w, model._stream())
  assert rc == 0, lib.tapir_last_error(ctx)
  torch.cuda.synchronize()
  from tapnet_amd.backbone import _same_pad
  xr = x.to(torch.bfloat16).float().permute(0, 3, 1, 2)
  ref = F.conv2d(_same_pad(xr, 7, 2), wt.to(torch.bfloat16).float().to(dev), stride=2).permute(0, 2, 3, 1)
  d = (y.float() - ref).abs()
  assert float(d.max()) < 1.6e-2 and float(d.mean()) < 1e-3, (float(d.max()), float(d.mean()))
  cnt = torch.tensor([min(rows.value, ho - t * rows.value) * wo for t in range(tiles.value)], device=dev,
                     dtype=torch.float64)
  pm, pM2 = part[..., 0].double(), part[..., 1].double()
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  gd = y.double()
  assert torch.allclose(tot_mean, gd.mean((1, 2)), atol=1e-5)
  assert torch.allclose(tot_M2 / (ho * wo), gd.var((1, 2), unbiased=False), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('n,h,w,cin,cout,ks,stride,shortcut', [
    (3, 64, 64, 64, 64, 3, 1, True), (2, 32, 32, 128, 128, 3, 1, False), (2, 32, 32, 256, 256, 3, 1, True),
    (2, 64, 64, 64, 128, 3, 2, False), (2, 32, 32, 128, 256, 1, 2, False), (2, 25, 31, 64, 128, 3, 2, False),
    (1, 128, 128, 64, 64, 3, 1, True)])
def test_conv_fused_f32_vs_torch(n, h, w, cin, cout, ks, stride, shortcut):
  """The f32 instantiation of the same kernel (exact-f32 MFMA; what the f32 parity build's backbone runs)
  against torch f32: no operand rounding, so the tolerance is 2e-4 absolute on values of a few units."""
  from tapnet_amd import tapir_model
  from tapnet_amd.backbone import _same_pad
  m32 = tapir_model.TAPIR(pyramid_level=1, extra_convs=False, weights=synthetic.make_weights(21, 1, False, backbone=False),
                          device='cuda:0', dtype='float32')
  lib, ctx = m32._lib, m32._ctx
  dev = m32.device
  g = torch.Generator(device='cpu').manual_seed(h + cin + ks)
  x = (torch.randn(n, h, w, cin, generator=g) * 1.5 + 0.5).to(dev)
  wt = (torch.randn(cout, cin, ks, ks, generator=g) / (ks * ks * cin) ** 0.5).contiguous()
  gamma = (torch.rand(cin, generator=g) + 0.5).to(dev)
  beta = (torch.randn(cin, generator=g) * 0.3).to(dev)
  ho, wo = -(-h // stride), -(-w // stride)
  sc = torch.randn(n, ho, wo, cout, generator=g).to(dev) if shortcut else None
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, h, w, cin, cout, ks, stride, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  part_in = torch.empty(n, 4, cin, 2, device=dev)
  stream = m32._stream()
  assert lib.tapir_inorm_stats(ctx, x.data_ptr(), None, None, part_in.data_ptr(), n, h * w, cin, 4, stream) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wt.data_ptr()), cout, cin, ks, ctypes.byref(ws)) == 0
  y = torch.zeros(n, ho, wo, cout, device=dev)
  part = torch.zeros(n, tiles.value, cout, 2, device=dev)
  ss = torch.empty(n, cin, 2, device=dev)
  rc = lib.tapir_conv_fused(ctx, x.data_ptr(), part_in.data_ptr(), 4, 0, gamma.data_ptr(), beta.data_ptr(),
                            ss.data_ptr(), ws, sc.data_ptr() if shortcut else None, y.data_ptr(), part.data_ptr(),
                            n, h, w, cin, cout, ks, stride, stream)
  assert rc == 0, lib.tapir_last_error(ctx)
  torch.cuda.synchronize()
  xd = x.double()
  mean = xd.mean((1, 2), keepdim=True)
  var = xd.var((1, 2), keepdim=True, unbiased=False)
  xn = torch.relu((xd - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double())
  ref = F.conv2d(_same_pad(xn.permute(0, 3, 1, 2), ks, stride), wt.double().to(dev), stride=stride).permute(0, 2, 3, 1)
  if shortcut:
    ref = ref + sc.double()
  d = (y.double() - ref).abs()
  assert float(d.max()) < 2e-4, float(d.max())
  gd = y.double()
  cnt = torch.tensor([min(rows.value, ho - t * rows.value) * wo for t in range(tiles.value)], device=dev,
                     dtype=torch.float64)
  pm, pM2 = part[..., 0].double(), part[..., 1].double()
  tot_mean = (pm * cnt[None, :, None]).sum(1) / cnt.sum()
  tot_M2 = (pM2 + cnt[None, :, None] * (pm - tot_mean[:, None]) ** 2).sum(1)
  assert torch.allclose(tot_mean, gd.mean((1, 2)), atol=1e-5)
  assert torch.allclose(tot_M2 / (ho * wo), gd.var((1, 2), unbiased=False), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('size', [256, 200])
def test_backbone_fused_convs_vs_miopen(model, size):
  """Backbone.features of the bf16 model with the HIP convolutions and with every convolution on MIOpen,
  both against the f32 backbone (MIOpen f32) of the same weights.  The grids are unit vectors per pixel;
  the gates are those of test_bf16_backbone_golden (cosine > 0.998, max component error 2.5e-2): the
  fused path has to be as close to f32 as the all-MIOpen bf16 path is (it rounds the residual add once
  instead of twice), and the two bf16 paths agree with each other to the same order."""
  from tapnet_amd import tapir_model
  bb = model._backbone
  m32 = tapir_model.TAPIR(pyramid_level=1, extra_convs=False, weights=synthetic.make_weights(21, 1, False),
                          device='cuda:0', dtype='float32')
  frames = torch.as_tensor(synthetic.make_video(3, 6, size, size), device=model.device)
  frames = frames.reshape(-1, size, size, 3).float()
  assert bb._wstream, 'the 3x3 kernels were not packed for the HIP convolution'
  m32._backbone.conv_mode = 'miopen'
  ref = [t.clone() for t in m32._backbone.features(frames)]
  m32._backbone.conv_mode = 'auto'       # the f32 build's own HIP convolutions against the library's
  assert m32._backbone._wstream
  for r, a in zip(ref, m32._backbone.features(frames)):
    assert float((r - a).abs().max()) < 2e-5, float((r - a).abs().max())
  bb.conv_mode = 'miopen'
  mio = [t.clone() for t in bb.features(frames)]
  bb.conv_mode = 'auto'
  fused = bb.features(frames)
  assert any(v is not None for v in bb._plans.values()), 'no shape ran on the HIP convolution'
  for name, r, a, b in (('lowres', ref[0], mio[0], fused[0]), ('hires', ref[1], mio[1], fused[1])):
    cos_m, cos_f, cos_mf = (r * a).sum(-1).min(), (r * b).sum(-1).min(), (a * b).sum(-1).min()
    err_m, err_f = (r - a).abs().max(), (r - b).abs().max()
    print(f'{name} {size}: min cos vs f32: miopen {float(cos_m):.5f} fused {float(cos_f):.5f}; fused vs miopen '
          f'{float(cos_mf):.5f}; max err vs f32: miopen {float(err_m):.4f} fused {float(err_f):.4f}')
    assert float(cos_f) > 0.998 and float(err_f) < 2.5e-2
    assert float(cos_f) > float(cos_m) - 5e-4     # no further from f32 than the MIOpen bf16 path
    assert float(cos_mf) > 0.997


def test_backbone_graph_replay_and_streams_match_eager(model):
  """Backbone.features: the hipGraph replay (third call with a shape on) and the frame groups on several
  streams return bit-for-bit what one eager launch sequence on one stream returns, for NEW frame contents
  (the replay copies them into its static input) -- every kernel is independent of how many frames a
  launch covers."""
  bb = model._backbone
  saved = (bb.graph_min_frames, bb.streams)
  try:
    f1 = torch.as_tensor(synthetic.make_video(5, 16, 128, 128), device=model.device).reshape(-1, 128, 128, 3).float()
    f2 = torch.as_tensor(synthetic.make_video(6, 16, 128, 128), device=model.device).reshape(-1, 128, 128, 3).float()
    bb.graph_min_frames, bb.streams = 0, 1
    ref = [t.clone() for t in bb.features(f2)]
    bb.graph_min_frames, bb.streams = 8, 2
    for _ in range(3):
      bb.features(f1)                       # the third call captures
    key_has_graph = any('graph' in e for e in bb._graphs.values())
    assert key_has_graph, 'no graph was captured'
    out = bb.features(f2)                   # replay with other frames
    assert all(torch.equal(a, b) for a, b in zip(out, ref))
    out1 = [t.clone() for t in bb.features(f1)]
    assert not torch.equal(out1[0], ref[0])
    bb.graph_min_frames, bb.streams = 0, 2   # eager, two groups of frames on two streams
    out = bb.features(f2)
    assert all(torch.equal(a, b) for a, b in zip(out, ref))
  finally:
    bb.graph_min_frames, bb.streams = saved


@pytest.mark.parametrize('dtype,n,h,w,small', [('float32', 2, 32, 32, 0), ('bfloat16', 3, 32, 32, 0), ('float32', 1, 64, 64, 0),
                                               ('bfloat16', 2, 64, 64, 0), ('bfloat16', 2, 24, 32, 0),
                                               # few-frame clips (tapir_conv_set_small): 256 -> 1024 in the form of csrc/conv_small.hpp
                                               ('bfloat16', 1, 32, 32, 1), ('bfloat16', 2, 24, 40, 1), ('bfloat16', 1, 64, 64, 1)])
def test_extra_convs_block_vs_torch(dtype, n, h, w, small):
  """One ExtraConvs block (tapir_model.py:159-186) through the C ABI -- tapir_layernorm_affine, tapir_xconv
  256 -> 1024 (+ bias + GELU), tapir_xconv 1024 -> 256 (+ bias + skip) -- against torch float64 on the same
  operands (bf16 build: operands and stored intermediates rounded to bf16 as the kernels round them)."""
  from tapnet_amd import tapir_model
  bf = dtype == 'bfloat16'
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=False, weights=synthetic.make_weights(21, 1, False, backbone=False),
                        device='cuda:0', dtype=dtype)
  lib, ctx, dev, st = m._lib, m._ctx, m.device, m._stream()
  assert lib.tapir_conv_set_small(ctx, small) == 0     # (a fresh context per case: nothing to restore)
  tt = torch.bfloat16 if bf else torch.float32
  g = torch.Generator(device='cpu').manual_seed(h + w + n)
  C = 256
  x = (torch.randn(n, h, w, C, generator=g) * 1.3 + 0.4).to(dev).to(tt)
  gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
  beta = (torch.randn(C, generator=g) * 0.2).to(dev)
  w1 = (torch.randn(4 * C, C, 3, 3, generator=g) / (9 * C) ** 0.5).contiguous()
  b1 = (torch.randn(4 * C, generator=g) * 0.1).to(dev)
  w2 = (torch.randn(C, 4 * C, 3, 3, generator=g) / (9 * 4 * C) ** 0.5).contiguous()
  b2 = (torch.randn(C, generator=g) * 0.1).to(dev)
  rows, tiles, c1, c2 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_xconv_plan(ctx, h, w, C, 4 * C, ctypes.byref(rows), ctypes.byref(tiles), ctypes.byref(c1)) == 0
  assert lib.tapir_xconv_plan(ctx, h, w, 4 * C, C, ctypes.byref(rows), ctypes.byref(tiles), ctypes.byref(c2)) == 0
  ws1, ws2 = ctypes.c_void_p(), ctypes.c_void_p()
  assert lib.tapir_xconv_pack(ctx, ctypes.c_void_p(w1.data_ptr()), 4 * C, C, c1.value, ctypes.byref(ws1)) == 0
  assert lib.tapir_xconv_pack(ctx, ctypes.c_void_p(w2.data_ptr()), C, 4 * C, c2.value, ctypes.byref(ws2)) == 0
  y = torch.zeros(n, h, w, C, device=dev, dtype=tt)
  r = torch.zeros(n, h, w, 4 * C, device=dev, dtype=tt)
  out = torch.zeros(n, h, w, C, device=dev, dtype=tt)
  assert lib.tapir_layernorm_affine(ctx, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), n * h * w, C, st) == 0
  assert lib.tapir_xconv(ctx, y.data_ptr(), ws1, b1.data_ptr(), None, r.data_ptr(), n, h, w, C, 4 * C, 1, st) == 0, lib.tapir_last_error(ctx)
  assert lib.tapir_xconv(ctx, r.data_ptr(), ws2, b2.data_ptr(), y.data_ptr(), out.data_ptr(), n, h, w, 4 * C, C, 0, st) == 0
  torch.cuda.synchronize()
  rd = (lambda t: t.to(torch.bfloat16).double()) if bf else (lambda t: t.double())
  yr = F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), eps=1e-5)
  tol = dict(atol=2e-2, rtol=1e-2) if bf else dict(atol=2e-4, rtol=0)
  torch.testing.assert_close(y.double(), yr, **tol)
  # each stage against the reference applied to the kernel's own (stored) input of that stage
  conv = lambda a, wt: F.conv2d(a.permute(0, 3, 1, 2), wt.to(dev), padding=1).permute(0, 2, 3, 1)
  rr = F.gelu(conv(y.double(), rd(w1)) + b1.double(), approximate='tanh')
  torch.testing.assert_close(r.double(), rr, **tol)
  orr = conv(r.double(), rd(w2)) + b2.double() + y.double()
  torch.testing.assert_close(out.double(), orr, **tol)
  if bf:
    assert float((r.double() - rr).abs().mean()) < 1.5e-3 and float((out.double() - orr).abs().mean()) < 3e-3
  for hnd in (ws1, ws2):
    assert lib.tapir_conv_free(ctx, hnd) == 0


def test_extra_convs_hip_vs_torch_path():
  """Backbone.features of the BootsTAPIR model with the ExtraConvs as HIP kernels and as MIOpen convolutions +
  torch glue (the round-2 path, kept as the A/B switch): f32 grids agree to 2e-5, bf16 grids are as close to
  the f32 ones either way."""
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(21, 1, True)
  frames = torch.as_tensor(synthetic.make_video(3, 6, 256, 256), device='cuda:0').reshape(-1, 256, 256, 3).float()
  m32 = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, weights=w, device='cuda:0', dtype='float32')
  bb = m32._backbone
  bb.extra_convs_mode = 'torch'
  ref = [t.clone() for t in bb.features(frames)]
  bb.extra_convs_mode = 'hip'
  got = bb.features(frames)
  assert bb._xstream, 'the ExtraConvs kernels were not packed'
  assert float((ref[0] - got[0]).abs().max()) < 2e-5, float((ref[0] - got[0]).abs().max())
  m16 = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, weights=w, device='cuda:0', dtype='bfloat16')
  b16 = m16._backbone
  b16.extra_convs_mode = 'torch'
  t16 = [t.clone() for t in b16.features(frames)]
  b16.extra_convs_mode = 'hip'
  h16 = b16.features(frames)
  cos_t, cos_h = (ref[0] * t16[0]).sum(-1).min(), (ref[0] * h16[0]).sum(-1).min()
  print(f'min cosine to the f32 grids: torch path {float(cos_t):.5f}, HIP path {float(cos_h):.5f}')
  assert float(cos_h) > 0.998 and float(cos_h) > float(cos_t) - 5e-4


def test_reloading_hot_path_weights_keeps_the_backbone():
  """A second load_weights / load_state_dict with hot-path weights only (no resnet_torch.* keys) re-runs
  tapir_finalize_weights; the backbone's packed weight streams -- and the hipGraphs that captured their addresses --
  belong to the Backbone object and must survive it (round-2 ADVICE: they were freed with the hot-path weights and
  the next get_feature_grids read freed memory).  A load WITH backbone weights rebuilds the backbone."""
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(21, 1, True)
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, weights=w, device='cuda:0', dtype='bfloat16')
  frames = torch.as_tensor(synthetic.make_video(3, 8, 128, 128), device='cuda:0')
  for _ in range(3):                                   # the third call captures the graph
    ref = [t.clone() for t in m.get_feature_grids(frames).lowres]
  bb = m._backbone
  assert any('graph' in e for e in bb._graphs.values())
  hot = {k: v for k, v in synthetic.make_weights(22, 1, True).items()
         if not (k.startswith('resnet_torch.') or k.startswith('extra_convs.'))}
  m.load_weights(hot)                                  # other mixer / head weights, same backbone
  assert m._backbone is bb
  junk = [torch.full((1 << 22,), 7.0, device='cuda:0') for _ in range(8)]   # reuse whatever was freed
  torch.cuda.synchronize()
  again = m.get_feature_grids(frames).lowres           # graph replay on the old packs
  assert all(torch.equal(a, b) for a, b in zip(again, ref))
  del junk
  m.load_state_dict(synthetic.make_weights(23, 1, True))   # new backbone weights: rebuilt, old packs released
  assert m._backbone is not bb and not bb._wstream and not bb._xstream
  other = m.get_feature_grids(frames).lowres
  assert not torch.equal(other[0], ref[0]) and torch.isfinite(other[0]).all()


@pytest.mark.parametrize('extra,size', [(False, 128), (True, 256)])
def test_chunked_feature_extraction_is_eager_and_identical(extra, size):
  """feature_extractor_chunk_size (tapir_model.py:689-703) bounds the backbone's scratch: chunked calls are launched
  eagerly (a captured graph would keep full-clip static buffers) and give the bits of the unchunked call, whatever
  the chunk length (every kernel is independent of the number of frames per launch; the convolution implementation is
  chosen from the whole clip, so a short last chunk runs the same kernels).  BootsTAPIR at 256 x 256: the 18-frame clip
  runs the 128-pixel form of the 256 -> 1024 ExtraConvs convolution (from 8 frames on, extra_convs.hpp) -- chunks of 5
  or 7 frames and a rank's frame shard (global_frames) must run it too, or their bits differ (the forms add the input
  channels in different orders)."""
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(21, 1, extra)
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=extra, weights=w, device='cuda:0', dtype='bfloat16')
  bb = m._backbone
  frames = torch.as_tensor(synthetic.make_video(4, 18, size, size), device='cuda:0').reshape(-1, size, size, 3).float()
  ref = [t.clone() for t in bb.features(frames)]
  keys = set(bb._graphs)
  for chunk in (16, 5, 7):                             # 16 + 2 frames: the last chunk is below hip_min_frames
    for _ in range(3):
      out = bb.features(frames, chunk)
    assert all(torch.equal(a, b) for a, b in zip(out, ref)), chunk
  assert set(bb._graphs) == keys                       # chunked calls never enter the graph cache
  if extra:
    assert bb._xplan(size // 8, size // 8, 256, 1024)[1] == 8 and bb._xplan(size // 8, size // 8, 1024, 256)[1] == 4
  for lo, hi in ((0, 9), (9, 18), (16, 18)):           # frame shards of the clip (tapnet_amd.distributed)
    part = bb.features(frames[lo:hi], global_frames=18)
    assert all(torch.equal(a, b[lo:hi]) for a, b in zip(part, ref)), (lo, hi)
  if extra:                                            # ... and on its own a 5-frame clip takes the small form
    bb.features(frames[:5])
    assert bb._xplan(size // 8, size // 8, 256, 1024)[1] == 4


@pytest.mark.parametrize('size,expect_hip', [(512, True), (576, True), (1040, False)])
def test_extra_convs_map_widths(size, expect_hip):
  """ExtraConvs on larger frames (f32 build): 512 x 512 -> a 64-wide low-res map (two output rows per tile in the wide
  form, input channels in chunks of 64: BASELINE configs[4]) and 576 x 576 -> 72 cells per row (one row per tile, wide
  form only) run the HIP kernels and agree with the MIOpen + torch path to 2e-5; 1040 x 1040 -> 130 cells per row is
  beyond the kernels (tapir_xconv_plan: TAPIR_ERR_UNSUPPORTED) and takes the library path -- silently equal, never
  wrong."""
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(21, 1, True)
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, weights=w, device='cuda:0', dtype='float32')
  bb = m._backbone
  frames = torch.as_tensor(synthetic.make_video(3, 4, size, size), device='cuda:0').reshape(-1, size, size, 3).float()
  bb.extra_convs_mode = 'torch'
  ref = [t.clone() for t in bb.features(frames)]
  bb.extra_convs_mode = 'hip'
  got = bb.features(frames)
  assert bool(bb._xstream) == expect_hip
  assert float((ref[0] - got[0]).abs().max()) < 2e-5, float((ref[0] - got[0]).abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', ['bfloat16', 'float32'])
def test_next_norm_merged_in_the_launch_is_bit_identical(dtype, monkeypatch):
  """The producing convolution merges the next InstanceNorm's (a, b) pairs itself (tapir_conv_fused_nn: write-through
  tile summaries, an arrival counter per image, the last workgroup merges) instead of a merge launch in front of the
  consumer: the feature grids are bit-identical to the separate launches -- eager and as a replayed hipGraph, twelve
  times over with other work in flight (a stale or torn summary would show as a changed pair) -- and the arrival
  counters are back at zero."""
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(5, 0, False)
  video = synthetic.make_video(7, 24, 256, 256)
  monkeypatch.setenv('TAPIR_FUSE_FINALIZE', '0')
  ref_model = tapir_model.TAPIR(pyramid_level=0, weights=w, dtype=dtype, device='cuda:0')
  assert not ref_model._backbone.fuse_finalize
  fg0 = ref_model.get_feature_grids(video, False)
  low0, hi0 = fg0.lowres[0].clone(), fg0.hires[0].clone()
  monkeypatch.setenv('TAPIR_FUSE_FINALIZE', '1')
  m = tapir_model.TAPIR(pyramid_level=0, weights=w, dtype=dtype, device='cuda:0')
  assert m._backbone.fuse_finalize
  noise = torch.randn(4096, 4096, device='cuda:0')
  for i in range(12):
    if i % 3 == 0:
      noise = noise @ noise.t() * 1e-4          # other kernels in flight around the backbone's streams
    fg = m.get_feature_grids(video, False)
    assert torch.equal(fg.lowres[0], low0) and torch.equal(fg.hires[0], hi0), i
  torch.cuda.synchronize()
  arrive = [t for k, t in m._backbone._bufs.items() if 'arrive' in k]
  assert arrive and all(int(t.abs().sum()) == 0 for t in arrive)


@pytest.mark.gpu
@pytest.mark.parametrize('dtype,size', [('bfloat16', 256), ('float32', 128), ('bfloat16', 200)])
def test_conv0_and_projection_in_one_launch_is_bit_identical(dtype, size, monkeypatch):
  """conv_0 + proj_conv of a group's first block as ONE launch (tapir_conv_fused_dual_nn, conv_fused.hpp DUAL; resnet.py:
  232-247) against the two launches (TAPIR_FUSE_PROJ=0): the feature grids of a whole backbone pass are bit-identical
  -- eager and as a replayed hipGraph -- in both builds, also on a frame size whose rows do not fill the pixel tiles
  (200 -> 100 -> 50 -> 25 cells).  (Odd inputs to the stride-2 pair -- centre tap (1, 1) -- cannot occur in the model:
  resolutions are multiples of 8; the emulator test covers them.)"""
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(6, 0, False)
  video = synthetic.make_video(8, 12, size, size)
  kw = dict(pyramid_level=0, weights=w, dtype=dtype, device='cuda:0', initial_resolution=(size, size))
  monkeypatch.setenv('TAPIR_FUSE_PROJ', '0')
  ref_model = tapir_model.TAPIR(**kw)
  assert not ref_model._backbone.fuse_proj
  fg0 = ref_model.get_feature_grids(video, False)
  low0, hi0 = fg0.lowres[0].clone(), fg0.hires[0].clone()
  monkeypatch.setenv('TAPIR_FUSE_PROJ', '1')
  m = tapir_model.TAPIR(**kw)
  assert m._backbone.fuse_proj and len(m._backbone._wdual) == 4      # groups 0-3: 64->64, 64->128 s2, 128->256 s2, 256->256
  assert 'one launch' in m._backbone.describe(12)
  for i in range(5):                                                   # (from the third call on: the captured graph, bf16)
    fg = m.get_feature_grids(video, False)
    assert torch.equal(fg.lowres[0], low0) and torch.equal(fg.hires[0], hi0), i


@pytest.mark.gpu
@pytest.mark.parametrize('n,h,w,shortcut', [(12, 32, 32, True), (48, 32, 32, False), (5, 32, 32, True), (7, 16, 16, True),
                                            (3, 24, 24, False)])
def test_flat_tiling_of_the_256_channel_convs_is_bit_identical(model, n, h, w, shortcut):
  """csrc/conv_flat.hpp (three consecutive 64-pixel slabs of the launch's (image, row) space per workgroup) against the
  per-image tiling (csrc/conv_fused.hpp) on the GPU: outputs, slab summaries and the in-launch merged (a, b) pairs of
  the next norm, bit for bit -- a frame group of the headline clip (12 x 32 x 32: 64 workgroups, every fifth or sixth
  straddling two images), the whole clip (256 workgroups), a ragged slab count, 16 x 16 maps (4-row slabs) and 24-wide
  rows (48-pixel slabs).  resnet.py:241-256; the per-image form is held to torch above."""
  lib, ctx = model._lib, model._ctx
  dev, stream, c = model.device, model._stream(), 256
  g = torch.Generator(device='cpu').manual_seed(n * 100 + h)
  x = (torch.randn(n, h, w, c, generator=g) * 1.5 + 0.5).to(torch.bfloat16).to(dev)
  sc = torch.randn(n, h, w, c, generator=g).to(torch.bfloat16).to(dev) if shortcut else None
  wt = (torch.randn(c, c, 3, 3, generator=g) / (9 * c) ** 0.5).contiguous()
  g0, b0 = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.3).to(dev)
  g1, b1 = (torch.rand(c, generator=g) + 0.5).to(dev), (torch.randn(c, generator=g) * 0.3).to(dev)
  rows, tiles = ctypes.c_int(), ctypes.c_int()
  assert lib.tapir_conv_plan(ctx, h, w, c, c, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
  part_in = torch.empty(n, 4, c, 2, device=dev)
  assert lib.tapir_inorm_stats(ctx, x.data_ptr(), None, None, part_in.data_ptr(), n, h * w, c, 4, stream) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wt.data_ptr()), c, c, 3, ctypes.byref(ws)) == 0
  from tapnet_amd import _ffi
  outs = []
  try:
    for mode in (0, 1):
      assert lib.tapir_debug_set_conv_flat(ctx, mode) == 0
      wgs = ctypes.c_int()
      rc = lib.tapir_conv_flat_plan(ctx, n, h, w, c, c, 3, 1, ctypes.byref(wgs))
      assert (rc == 0 and wgs.value == -(-n * tiles.value // 3)) if mode else rc == _ffi.TAPIR_ERR_UNSUPPORTED
      for rep in range(3):     # (repeats: a torn or stale summary would change a pair)
        y = torch.zeros(n, h, w, c, device=dev, dtype=torch.bfloat16)
        part = torch.full((n, tiles.value, c, 2), float('nan'), device=dev)
        ss = torch.empty(n, c, 2, device=dev)
        ssn = torch.full((n, c, 2), float('nan'), device=dev)
        arrive = torch.zeros(n, dtype=torch.int32, device=dev)
        nn = _ffi.TapirNextNorm(g1.data_ptr(), b1.data_ptr(), ssn.data_ptr(), arrive.data_ptr())
        rc = lib.tapir_conv_fused_nn(ctx, x.data_ptr(), part_in.data_ptr(), 4, 0, g0.data_ptr(), b0.data_ptr(), ss.data_ptr(),
                                     ws, sc.data_ptr() if shortcut else None, y.data_ptr(), part.data_ptr(), n, h, w, c, c,
                                     3, 1, ctypes.byref(nn), stream)
        assert rc == 0, lib.tapir_last_error(ctx)
        torch.cuda.synchronize()
        assert int(arrive.abs().sum()) == 0
        assert torch.isfinite(part).all() and torch.isfinite(ssn).all()
        outs.append((y, part, ssn))
  finally:
    assert lib.tapir_debug_set_conv_flat(ctx, 0) == 0   # (the default)
    lib.tapir_conv_free(ctx, ws)
  for y, part, ssn in outs[1:]:
    assert torch.equal(y, outs[0][0]) and torch.equal(part, outs[0][1]) and torch.equal(ssn, outs[0][2])



@pytest.mark.gpu
@pytest.mark.parametrize('n,h,w,cin,cout,ks,stride,shortcut', [
    (1, 128, 128, 64, 64, 3, 1, True), (1, 64, 64, 128, 128, 3, 1, True), (1, 32, 32, 256, 256, 3, 1, True),
    (3, 32, 32, 256, 256, 3, 1, False), (1, 128, 128, 64, 128, 3, 2, False), (1, 64, 64, 128, 256, 3, 2, False),
    (1, 128, 128, 64, 128, 1, 2, False), (1, 32, 32, 256, 256, 1, 1, False), (2, 24, 40, 256, 256, 3, 1, True)])
def test_few_frame_form_of_the_block_convs_matches_the_many_frame_kernel(model, n, h, w, cin, cout, ks, stride, shortcut):
  """csrc/conv_small.hpp (a workgroup per (row tile, 64 output channels), the waves split the taps) against
  csrc/conv_fused.hpp on the GPU, the layers of ONE 256 x 256 frame (the online model, tapnet/live_demo.py:51-77) and a
  non-square map: outputs equal up to the summation order (a bf16 step on a few values), the in-launch merged (a, b) pairs
  of the next norm agree to f32 noise, counters back at zero, repeatable bit for bit.  resnet.py:241-256."""
  lib, ctx = model._lib, model._ctx
  dev, stream = model.device, model._stream()
  g = torch.Generator(device='cpu').manual_seed(n * 100 + h + cin + ks)
  ho, wo = -(-h // stride), -(-w // stride)
  x = (torch.randn(n, h, w, cin, generator=g) * 1.5 + 0.5).to(torch.bfloat16).to(dev)
  sc = torch.randn(n, ho, wo, cout, generator=g).to(torch.bfloat16).to(dev) if shortcut else None
  wt = (torch.randn(cout, cin, ks, ks, generator=g) / (ks * ks * cin) ** 0.5).contiguous()
  g0, b0 = (torch.rand(cin, generator=g) + 0.5).to(dev), (torch.randn(cin, generator=g) * 0.3).to(dev)
  g1, b1 = (torch.rand(cout, generator=g) + 0.5).to(dev), (torch.randn(cout, generator=g) * 0.3).to(dev)
  part_in = torch.empty(n, 4, cin, 2, device=dev)
  assert lib.tapir_inorm_stats(ctx, x.data_ptr(), None, None, part_in.data_ptr(), n, h * w, cin, 4, stream) == 0
  ws = ctypes.c_void_p()
  assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wt.data_ptr()), cout, cin, ks, ctypes.byref(ws)) == 0
  from tapnet_amd import _ffi
  res = {}
  try:
    for small in (0, 1):
      assert lib.tapir_conv_set_small(ctx, small) == 0
      rows, tiles = ctypes.c_int(), ctypes.c_int()
      assert lib.tapir_conv_plan(ctx, h, w, cin, cout, ks, stride, ctypes.byref(rows), ctypes.byref(tiles)) == 0
      outs = []
      for rep in range(3):
        y = torch.zeros(n, ho, wo, cout, device=dev, dtype=torch.bfloat16)
        part = torch.full((n, tiles.value, cout, 2), float('nan'), device=dev)
        ss = torch.empty(n, cin, 2, device=dev)
        ssn = torch.full((n, cout, 2), float('nan'), device=dev)
        arrive = torch.zeros(n, dtype=torch.int32, device=dev)
        nn = _ffi.TapirNextNorm(g1.data_ptr(), b1.data_ptr(), ssn.data_ptr(), arrive.data_ptr())
        rc = lib.tapir_conv_fused_nn(ctx, x.data_ptr(), part_in.data_ptr(), 4, 0, g0.data_ptr(), b0.data_ptr(), ss.data_ptr(),
                                     ws, sc.data_ptr() if shortcut else None, y.data_ptr(), part.data_ptr(), n, h, w, cin, cout,
                                     ks, stride, ctypes.byref(nn), stream)
        assert rc == 0, lib.tapir_last_error(ctx)
        torch.cuda.synchronize()
        assert int(arrive.abs().sum()) == 0 and torch.isfinite(part).all() and torch.isfinite(ssn).all()
        outs.append((y, part, ssn))
      for y, part, ssn in outs[1:]:
        assert torch.equal(y, outs[0][0]) and torch.equal(part, outs[0][1]) and torch.equal(ssn, outs[0][2])
      res[small] = outs[0] + (rows.value * wo, tiles.value)
  finally:
    assert lib.tapir_conv_set_small(ctx, 0) == 0
    lib.tapir_conv_free(ctx, ws)
  assert res[1][3] <= 128 and res[1][4] >= res[0][4]
  yb, ys = res[0][0].float(), res[1][0].float()
  d = (yb - ys).abs()
  assert float(d.max()) <= 2e-2 * max(1.0, float(yb.abs().max())) and float((d > 0).float().mean()) < 0.05, (float(d.max()), float((d > 0).float().mean()))
  # the merged pairs: a = rstd * gamma, b = beta - mean * a of (nearly) the same stored tensor
  torch.testing.assert_close(res[1][2], res[0][2], rtol=2e-3, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('size,frames', [(256, 1), (256, 3), (512, 1), (192, 2)])
def test_few_frame_backbone_matches_the_library_path(size, frames):
  """Backbone.features of a clip of fewer than 4 frames (the online model's launches, tapnet/live_demo.py:51-77) with the
  few-frame HIP convolutions (csrc/conv_small.hpp: every ResNet block convolution and both ExtraConvs convolutions) against
  the path of rounds 2-5 (MIOpen / CK convolutions + torch glue + HIP norm kernels): the L2-normalised bf16 grids agree
  per cell (cosine >= 0.995: two bf16 implementations of 29 convolutions), also where single layers fall back to the
  many-frame kernel (512 x 512: output rows longer than 128 pixels) and on maps that are no multiple of the tile."""
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(23, 1, True)
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, weights=w, device='cuda:0', dtype='bfloat16')
  bb = m._backbone
  fr = torch.as_tensor(synthetic.make_video(5, frames, size, size), device='cuda:0').reshape(-1, size, size, 3).float()
  outs = {}
  for small in (False, True):
    bb.small_convs = small
    low, hi = bb.features(fr)
    torch.cuda.synchronize()
    assert torch.isfinite(low).all() and torch.isfinite(hi).all()
    assert bb._small_now == small
    outs[small] = (low.clone(), hi.clone())
    if small:                            # repeatable bit for bit (the library's convolutions accumulate with atomics and are not)
      low2, hi2 = bb.features(fr)
      assert torch.equal(low2, outs[small][0]) and torch.equal(hi2, outs[small][1])
  for a, b in zip(outs[False], outs[True]):
    cos = (a * b).sum(-1)                # both are unit vectors per cell
    assert float(cos.min()) >= 0.995, float(cos.min())
  # and a many-frame clip afterwards takes the many-frame kernels again
  bb.features(torch.cat([fr] * 4)[:8] if frames * 4 >= 8 else torch.cat([fr] * 8)[:8])
  assert not bb._small_now

@pytest.mark.gpu
def test_backbone_with_the_flat_tiling_is_bit_identical(monkeypatch):
  """A whole backbone pass (eager and the replayed hipGraph, four frame groups on four streams) with the 256-channel
  convolutions of ResNet groups 2 and 3 in the flat tiling (incl. the dual conv_0 + proj_conv launch of group 3) against
  TAPIR_CONV_FLAT=0: bit-identical feature grids, and a 5-frame shard of the clip equals the clip's first 5 frames."""
  from tapnet_amd import tapir_model
  w = synthetic.make_weights(9, 0, False)
  video = synthetic.make_video(10, 24, 256, 256)
  kw = dict(pyramid_level=0, weights=w, dtype='bfloat16', device='cuda:0')
  monkeypatch.setenv('TAPIR_CONV_FLAT', '0')
  ref_model = tapir_model.TAPIR(**kw)
  fg0 = ref_model.get_feature_grids(video, False)
  low0, hi0 = fg0.lowres[0].clone(), fg0.hires[0].clone()
  monkeypatch.setenv('TAPIR_CONV_FLAT', '1')
  m = tapir_model.TAPIR(**kw)
  wgs = ctypes.c_int()
  assert m._lib.tapir_conv_flat_plan(m._ctx, 6, 32, 32, 256, 256, 3, 1, ctypes.byref(wgs)) == 0 and wgs.value == 32
  for i in range(5):
    fg = m.get_feature_grids(video, False)
    assert torch.equal(fg.lowres[0], low0) and torch.equal(fg.hires[0], hi0), i
  fg5 = m.get_feature_grids(video[:, :5], False)
  assert torch.equal(fg5.lowres[0], low0[:, :5]) and torch.equal(fg5.hires[0], hi0[:, :5])
