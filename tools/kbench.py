#!/usr/bin/env python
"""Kernel micro-benchmarks on one MI355X: every hot kernel of the TAPIR path at the
config-2 shapes (256 queries x 48 frames = 12288 token rows), timed with events on the
launch stream, checked against a torch reference, reported against the roofline.

    python tools/kbench.py [--what gemm,mix,mixer,backbone,norm,gemmtrace,gemmsteps,mixtrace,cvtrace]
                           [--shapes up,down,up4,...] [--tiles 3,8,14,...] [--reps 40] [--out gpurun_out/kbench.json]

--what gemmtrace / gemmsteps / mixtrace / cvtrace print in-kernel phase traces (wall-clock stamps or
per-wave cycle totals written by the TRACE builds of the kernels), norm the backbone glue kernels.

GEMM rows: every tile shape of gemm.hpp (tapir_debug_gemm hook) on the mixer shapes
(up: [R,512]x[512,2048]+GELU -> bf16; down: [R,2048]x[2048,512]+skip -> f32), the first /
last linear and the cost-volume einsum.  Buffers rotate over three sets so that inputs are
not trivially L2-resident.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np
import torch

from tapnet_amd import _ffi, synthetic, tapir_model

TILES = {1: '192x128s4', 2: '128x128', 3: '192x64', 4: '192x128s3', 5: '192x128ws', 6: '192x256', 7: '256x128s3', 8: '128x128w8', 9: '192x64w8', 10: '128x64w8', 11: '256x128w16pf', 12: '128x128w8pf', 13: '256x128w16', 14: '256x128w16s3', 15: '128x128w8s3', 16: '256x256'}
R = 256 * 48


def timeit(fn, reps, warm=5):
  for _ in range(warm):
    fn(0)
  torch.cuda.synchronize()
  evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
  for i, (a, b) in enumerate(evs):
    a.record()
    fn(i)
    b.record()
  torch.cuda.synchronize()
  t = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
  return dict(med_us=round(t[len(t) // 2], 2), min_us=round(t[0], 2), p90_us=round(t[int(len(t) * 0.9)], 2))


def timeit_batch(fn, reps, warm=3):
  """back-to-back launches between two events (no per-launch event overhead)"""
  for _ in range(warm):
    fn(0)
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for i in range(reps):
    fn(i)
  b.record()
  torch.cuda.synchronize()
  return round(a.elapsed_time(b) * 1e3 / reps, 2)


def gelu_tanh(x):
  return torch.nn.functional.gelu(x, approximate='tanh')


def bench_gemm(model, reps, results, only_shapes=None, only_tiles=None):
  lib, ctx = model._lib, model._ctx
  bf = model.dtype == 'bfloat16'
  tdt = torch.bfloat16 if bf else torch.float32
  dev = model.device
  stream = model._stream()
  shapes = [('up', R, 2048, 512, 1), ('down', R, 512, 2048, 2), ('first', R, 512, 576, 0),
            ('last', R, 388, 512, 0), ('costvol', 256, 49152, 256, 0),
            # the online model's per-frame shapes (256 points x 1 frame)
            ('up256', 256, 2048, 512, 1), ('down256', 256, 512, 2048, 2), ('costvol1', 256, 1024, 256, 0),
            # one rank's share of config 3 (1024 queries x 48 frames)
            ('up4', 4 * R, 2048, 512, 1), ('down4', 4 * R, 512, 2048, 2),
            # config 5's cost volume as a GEMM writing the f32 volume (row N3): one 256-MiB query chunk, and all 4096 queries
            ('costvol5', 682, 98304, 256, 0), ('costvol5full', 4096, 98304, 256, 0)]
  if not only_shapes:
    shapes = [s_ for s_ in shapes if not s_[0].startswith('costvol5')]   # (1.6-GB outputs: only when asked for)
  g = torch.Generator(device='cpu').manual_seed(0)
  for name, M, N, K, epi in shapes:
    nset = 3
    A = [(torch.randn(M, K, generator=g) * 1.0).to(dev, tdt) for _ in range(nset)]
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev, tdt)
    bias = torch.randn(N, generator=g).to(dev)
    resid = torch.randn(M, N, generator=g).to(dev) if epi == 2 else None
    out_dt = tdt if epi == 1 else torch.float32
    C = [torch.empty(M, N, device=dev, dtype=out_dt) for _ in range(nset)]
    ref = A[0].float() @ W.float().t() + bias
    if epi == 1:
      ref = gelu_tanh(ref)
    if epi == 2:
      ref = ref + resid
    flops = 2.0 * M * N * K
    es = 2 if bf else 4
    bytes_alg = M * K * es + N * K * es + M * N * (es if epi == 1 else 4) + (M * N * 4 if epi == 2 else 0)
    if only_shapes and name not in only_shapes:
      continue
    variants = [(t, n) for t, n in TILES.items() if not only_tiles or t in only_tiles]
    for tile, tname in variants:
      def run(i, tile=tile):
        k = i % nset
        rc = lib.tapir_debug_gemm(ctx, A[k].data_ptr(), K, W.data_ptr(), K, bias.data_ptr(),
                                  resid.data_ptr() if resid is not None else None, N,
                                  C[k].data_ptr(), N, M, N, K, epi, tile, stream)
        assert rc == 0, lib.tapir_last_error(ctx)
      run(0)
      torch.cuda.synchronize()
      err = float((C[0].float() - ref).abs().max())
      t = timeit(run, reps)
      tb = timeit_batch(run, reps)
      row = dict(kernel=f'gemm_{name}', tile=tname, M=M, N=N, K=K, dtype=model.dtype, max_err=round(err, 5),
                 **t, batch_us=tb, tflops=round(flops / (t['med_us'] * 1e-6) / 1e12, 1),
                 out_GBps=round(M * N * (es if epi == 1 else 4) / (t['med_us'] * 1e-6) / 1e9, 1),
                 frac_of_mfma_peak=round(flops / (t['med_us'] * 1e-6) / (2.5e15 if bf else 157.3e12), 4),
                 tflops_batch=round(flops / (tb * 1e-6) / 1e12, 1),
                 alg_GBps=round(bytes_alg / (t['med_us'] * 1e-6) / 1e9, 1))
      results.append(row)
      print(json.dumps(row), flush=True)
    del A, C


def bench_fill(model, reps, results):
  """Write-only and copy ceilings of this box (the rooflines of the kernels that are bound by what they store): torch
  fill_ / copy_ of 256-MiB and 1.5-GiB f32 buffers, and the library GEMM at the cost-volume shape with a bf16 output."""
  dev = model.device
  for mb in (256, 1536):
    a = torch.empty(mb << 18, device=dev)
    b = torch.empty_like(a)
    t = timeit(lambda i: a.fill_(1.0), max(5, reps // 2), warm=2)
    row = dict(kernel='fill_f32', MiB=mb, **t, write_GBps=round(a.numel() * 4 / (t['med_us'] * 1e-6) / 1e9, 1))
    results.append(row); print(json.dumps(row), flush=True)
    t = timeit(lambda i: b.copy_(a), max(5, reps // 2), warm=2)
    row = dict(kernel='copy_f32', MiB=mb, **t, write_GBps=round(a.numel() * 4 / (t['med_us'] * 1e-6) / 1e9, 1),
               read_plus_write_GBps=round(2 * a.numel() * 4 / (t['med_us'] * 1e-6) / 1e9, 1))
    results.append(row); print(json.dumps(row), flush=True)
    del a, b
  for M in (682, 4096):
    A = torch.randn(M, 256, device=dev, dtype=torch.bfloat16)
    W = torch.randn(98304, 256, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, 98304, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda i: torch.mm(A, W.t(), out=out), max(5, reps // 2), warm=2)
    f = 2.0 * M * 98304 * 256
    row = dict(kernel='library_gemm_bf16_out', M=M, N=98304, K=256, **t, tflops=round(f / (t['med_us'] * 1e-6) / 1e12, 1),
               frac_of_mfma_peak=round(f / (t['med_us'] * 1e-6) / 2.5e15, 4),
               out_GBps=round(M * 98304 * 2 / (t['med_us'] * 1e-6) / 1e9, 1))
    results.append(row); print(json.dumps(row), flush=True)


def bench_mix(model, reps, results):
  lib, ctx = model._lib, model._ctx
  bf = model.dtype == 'bfloat16'
  dev = model.device
  stream = model._stream()
  N, T = 256, 48
  x = [torch.randn(N, T, 512, device=dev) for _ in range(3)]
  xo = [torch.empty(N, T, 512, device=dev) for _ in range(3)]
  xn = [torch.empty(N * T, 512, device=dev, dtype=torch.bfloat16 if bf else torch.float32) for _ in range(3)]

  for tc in (0, 512, 1024):
   def run(i, tc=tc):
    k = i % 3
    rc = lib.tapir_debug_mix(ctx, 0, x[k].data_ptr(), xo[k].data_ptr(), xn[k].data_ptr(), N, T, tc, stream)
    assert rc == 0, lib.tapir_last_error(ctx)
   t = timeit(run, reps)
   tb = timeit_batch(run, reps)
   byts = N * T * 512 * (4 + 4 + (2 if bf else 4))
   row = dict(kernel='mix', tile=f'tc{tc}', dtype=model.dtype, **t, batch_us=tb,
              alg_GBps=round(byts / (t['med_us'] * 1e-6) / 1e9, 1), alg_bytes=byts)
   results.append(row)
   print(json.dumps(row), flush=True)


def trace_mix(model):
  """per-unit phase stamps of the mix kernel (100 MHz wall clock): where does the time go"""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  N, T = 256, 48
  units = N * 4
  x = torch.randn(N, T, 512, device=dev); xo = torch.empty_like(x)
  xn = torch.empty(N * T, 512, device=dev, dtype=torch.bfloat16 if model.dtype == 'bfloat16' else torch.float32)
  tr = torch.zeros(units, 6, dtype=torch.int64, device=dev)
  for tc in (0, 1024):
    for _ in range(3):
      lib.tapir_debug_mix(ctx, 0, x.data_ptr(), xo.data_ptr(), xn.data_ptr(), N, T, tc, model._stream())
    torch.cuda.synchronize()
    lib.tapir_debug_set_trace(ctx, tr.data_ptr())
    lib.tapir_debug_mix(ctx, 0, x.data_ptr(), xo.data_ptr(), xn.data_ptr(), N, T, tc, model._stream())
    torch.cuda.synchronize()
    lib.tapir_debug_set_trace(ctx, None)
    t = tr.cpu().numpy().astype(np.float64) * 0.01   # us
    t0 = t[:, 0].min()
    t -= t0
    names = ['stage issued', 'rows landed', 'process start', 'stats done', 'stream done', 'stores issued']
    print(f'mix trace grid cap {tc}: kernel span {t.max():.1f} us')
    for k in range(6):
      print(f'  {names[k]:14s} min {t[:, k].min():6.1f}  median {np.median(t[:, k]):6.1f}  max {t[:, k].max():6.1f}')
    d = np.diff(t, axis=1)
    for k in range(5):
      print(f'  phase {names[k]} -> {names[k + 1]}: median {np.median(d[:, k]):6.2f} us  p90 {np.percentile(d[:, k], 90):6.2f}')


def trace_gemm(model):
  """per-workgroup stamps of the mixer GEMMs: prologue, k-loop and epilogue of every tile"""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  tdt = torch.bfloat16
  for name, M, N, K, epi, tile in (('up', R, 2048, 512, 1, 8), ('up', R, 2048, 512, 1, 1), ('up', R, 2048, 512, 1, 6),
                                   ('up', R, 2048, 512, 1, 2), ('down', R, 512, 2048, 2, 3), ('down', R, 512, 2048, 2, 1),
                                   ('down', R, 512, 2048, 2, 7)):
    A = torch.randn(M, K, device=dev).to(tdt); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(tdt)
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=tdt if epi == 1 else torch.float32)
    tr = torch.zeros(512, 16, dtype=torch.int64, device=dev)
    def run():
      lib.tapir_debug_gemm(ctx, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), resid.data_ptr(), N,
                           C.data_ptr(), N, M, N, K, epi, tile, model._stream())
    for _ in range(3):
      run()
    torch.cuda.synchronize()
    lib.tapir_debug_set_trace(ctx, tr.data_ptr()); run(); torch.cuda.synchronize()
    lib.tapir_debug_set_trace(ctx, None)
    t = tr.cpu().numpy().astype(np.float64) * 0.01
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    nst = int((t[0] > 0).sum())
    print(f'gemm {name} tile {TILES[tile]}: {t.shape[0]} workgroups, {nst} stamps, span {(t[:, :nst].max() - t0):.1f} us')
    print('  start spread (us): median', round(float(np.median(t[:, 0] - t0)), 2), 'max', round(float((t[:, 0] - t0).max()), 2))
    d = np.diff(t[:, :nst], axis=1)
    labels = ['prologue (first stage landed)'] + sum([[f'k-loop tile {i}', f'epilogue tile {i}'] for i in range(8)], [])
    for k in range(nst - 1):
      print(f'  {labels[k]:32s} median {np.median(d[:, k]):6.2f} us   p90 {np.percentile(d[:, k], 90):6.2f}')


def trace_gemm_steps(model):
  """where a k-step goes: per-wave shader-cycle totals (s_memtime) of copy issue / fragment reads +
  MFMAs / wait for own copies / barrier / epilogue, from the TRACE build of gemm_nt_kernel"""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  tdt = torch.bfloat16
  for name, M, N, K, epi, tile in (('up', R, 2048, 512, 1, 8), ('up', R, 2048, 512, 1, 6), ('up', R, 2048, 512, 1, 1),
                                   ('down', R, 512, 2048, 2, 3), ('down', R, 512, 2048, 2, 8), ('down', R, 512, 2048, 2, 1)):
    A = torch.randn(M, K, device=dev).to(tdt); W = (torch.randn(N, K, device=dev) / K ** 0.5).to(tdt)
    bias = torch.randn(N, device=dev); resid = torch.randn(M, N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=tdt if epi == 1 else torch.float32)
    tr = torch.zeros(768 * 16, 8, dtype=torch.int64, device=dev)
    lib.tapir_debug_set_trace(ctx, tr.data_ptr())
    def run(flag):
      rc = lib.tapir_debug_gemm(ctx, A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), resid.data_ptr(), N,
                                C.data_ptr(), N, M, N, K, epi, tile | flag, model._stream())
      assert rc == 0, lib.tapir_last_error(ctx)
    for _ in range(3):
      run(1 << 20)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tr.zero_()
    e0.record(); run(1 << 20); e1.record(); torch.cuda.synchronize()
    traced_us = e0.elapsed_time(e1) * 1e3
    lib.tapir_debug_set_trace(ctx, None)
    for _ in range(3):
      run(0)
    e0.record()
    for _ in range(10):
      run(0)
    e1.record(); torch.cuda.synchronize()
    plain_us = e0.elapsed_time(e1) * 1e2
    t = tr.cpu().numpy().astype(np.float64)
    t = t[t[:, 5] > 0]
    tot = t[:, 5]
    mhz = np.median(tot) / traced_us   # cycles per us of the traced run, if s_memtime counts shader cycles
    print(f'gemm {name} tile {TILES[tile]}: plain {plain_us:.1f} us, traced {traced_us:.1f} us, {t.shape[0]} waves, '
          f'median wave lifetime {np.median(tot):.0f} cycles (~{mhz:.0f} MHz if it spans the launch)')
    for k, lab in enumerate(['copy issue', 'frag reads + MFMA', 'wait own copies (vmcnt)', 'barrier', 'epilogue + next acc init']):
      print(f'  {lab:28s} median {np.median(t[:, k]):9.0f} cyc  {100 * np.median(t[:, k] / tot):5.1f} %   '
            f'p10 {100 * np.percentile(t[:, k] / tot, 10):5.1f} %  p90 {100 * np.percentile(t[:, k] / tot, 90):5.1f} %')
    rest = tot - t[:, :5].sum(axis=1)
    print(f'  {"prologue / other":28s} median {np.median(rest):9.0f} cyc  {100 * np.median(rest / tot):5.1f} %')


def trace_cv(model):
  """phase totals of cv_heads_mfma_kernel per workgroup (wall clock, 100 MHz), config-2 cost-volume stage"""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  B, Q, T, h, w = 1, 256, 48, 32, 32
  qf = torch.nn.functional.normalize(torch.randn(B, Q, 256, device=dev), dim=-1)
  grid = torch.nn.functional.normalize(torch.randn(B, T, h, w, 256, device=dev), dim=-1)
  pts = torch.empty(B, Q, T, 2, device=dev); occ = torch.empty(B, Q, T, device=dev); ex = torch.empty(B, Q, T, device=dev)
  tr = torch.zeros(512, 8, dtype=torch.int64, device=dev)
  def run():
    rc = lib.tapir_tracks_from_cost_volume(ctx, qf.data_ptr(), grid.data_ptr(), None, B, Q, T, h, w,
                                           pts.data_ptr(), occ.data_ptr(), ex.data_ptr(), model._stream())
    assert rc == 0, lib.tapir_last_error(ctx)
  for _ in range(3):
    run()
  torch.cuda.synchronize()
  lib.tapir_debug_set_trace(ctx, tr.data_ptr()); run(); torch.cuda.synchronize()
  lib.tapir_debug_set_trace(ctx, None)
  t = tr.cpu().numpy().astype(np.float64) * 0.01   # us per workgroup (24 maps each)
  names = ['barrier (previous map done)', 'stage cost map + prefetch + barrier', 'conv 1->16 + relu (+barrier)',
           'conv 16->1', 'argmax + softmax sums', 'occlusion conv (MFMA) + barrier', 'tail (one wave)']
  tot = t[:, :7].sum(axis=1)
  print(f'cv_heads trace: {t.shape[0]} workgroups, median total {np.median(tot):.1f} us for {Q * T // 512} maps each')
  for k in range(7):
    print(f'  {names[k]:40s} median {np.median(t[:, k]):7.1f} us  {100 * np.median(t[:, k] / tot):5.1f} %')


def bench_norm(model, reps, results):
  """backbone glue kernels (HBM-bound): InstanceNorm statistics (with / without the fused residual add)
  and normalise + ReLU at the ResNet's activation shapes, over slab counts"""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  stream = model._stream()
  for (n, h, w, c) in ((48, 128, 128, 64), (48, 64, 64, 128), (48, 32, 32, 256)):
    x = [torch.randn(n, h, w, c, device=dev).to(torch.bfloat16) for _ in range(3)]
    b = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    y = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    gamma = torch.ones(c, device=dev); beta = torch.zeros(c, device=dev)
    ss = torch.empty(n, c, 2, device=dev)
    nbytes = n * h * w * c * 2
    for slabs in (11, 22, 43, 64, 128):
      if slabs > h * w // 64:
        continue
      part = torch.empty(n, slabs, c, 2, device=dev)
      def stats(i):
        assert lib.tapir_inorm_stats(ctx, x[i % 3].data_ptr(), None, None, part.data_ptr(), n, h * w, c, slabs, stream) == 0
      def stats_add(i):
        assert lib.tapir_inorm_stats(ctx, x[i % 3].data_ptr(), b.data_ptr(), x[i % 3].data_ptr(), part.data_ptr(), n, h * w, c, slabs, stream) == 0
      def relu(i):
        assert lib.tapir_inorm_relu(ctx, x[i % 3].data_ptr(), part.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                    ss.data_ptr(), y.data_ptr(), None, n, h, w, c, slabs, 0, h, w, stream) == 0
      for name, fn, passes in (('inorm_stats', stats, 1), ('inorm_stats_add', stats_add, 3), ('inorm_relu(+finalize)', relu, 2)):
        t = timeit(fn, reps)
        row = dict(kernel=name, shape=[n, h, w, c], slabs=slabs, **t,
                   GBps=round(passes * nbytes / (t['med_us'] * 1e-6) / 1e9, 1))
        results.append(row)
        print(json.dumps(row), flush=True)


def bench_mixer(model, reps, results, shapes=((256, 48), (1024, 48), (512, 48), (384, 48), (128, 48), (256, 24), (256, 40), (256, 96), (1024, 96))):
  """whole PIPSMLPMixer (12 blocks) through the public C ABI (tapir_pips_mixer: staging of the input
  rows + mixer + copy of the result), A/B: separate launches (mode 1) vs the track-resident fused
  kernel (mode 2), same inputs; max |fused - separate| is the cross-check."""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  cin = 388 + 49 * (2 + model.pyramid_level)
  stream = model._stream()
  for N, T in shapes:
    x = torch.randn(N, T, cin, device=dev)
    outs = {}
    modes = [(1, 'separate')]
    if T <= 48:
      modes.append((2, 'fused'))
    if model.dtype == 'bfloat16' and T > 16:
      modes.append((3, 'fused_wide'))
    if os.environ.get('KBENCH_PAIRSIM') and model.dtype == 'bfloat16' and T == 48 and N % 2 == 0:
      modes.append((4, 'pair_sim_TIMING_ONLY'))   # (-DTAPIR_EXPERIMENTS builds; meaningless outputs)
    for mode, name in modes:
      assert lib.tapir_debug_set_mixer_mode(ctx, mode) == 0
      out = torch.empty(N, T, 388, device=dev)

      def run(i):
        rc = lib.tapir_pips_mixer(ctx, x.data_ptr(), N, T, out.data_ptr(), None, None, None, None, stream)
        assert rc == 0, lib.tapir_last_error(ctx)
      t = timeit(run, max(5, reps // 4), warm=2)
      outs[name] = out.clone()
      flops = 2.0 * N * T * (cin * 512 + 12 * 2 * 512 * 2048 + 512 * 388)
      row = dict(kernel=f'pips_mixer_12blocks_{name}', N=N, T=T, dtype=model.dtype, **t,
                 tflops=round(flops / (t['med_us'] * 1e-6) / 1e12, 1),
                 us_per_block=round(t['med_us'] / 12, 2))
      if name != 'separate':
        d = (outs[name] - outs['separate']).abs()
        row['max_abs_diff_vs_separate'] = float(d.max())
        row['median_abs_diff_vs_separate'] = float(d.median())
        row['finite'] = bool(torch.isfinite(outs[name]).all())
      results.append(row)
      print(json.dumps(row), flush=True)
    assert lib.tapir_debug_set_mixer_mode(ctx, 0) == 0


def bench_cv(model, reps, results):
  """cost-volume stage (tapir_tracks_from_cost_volume: casts + einsum + heads) at config 2 and at one
  rank's share of config 3, A/B: mode 1 = einsum GEMM into a workspace + heads kernel, mode 2 = the
  fused kernel in its pixel-tiled form (costvol_fused.hpp, no volume in HBM), mode 0 = automatic = the
  row-streamed fused kernel (costvol_rows.hpp: every wave owns whole maps)."""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  stream = model._stream()
  g = torch.Generator(device='cpu').manual_seed(0)
  for Q, T, H in ((256, 48, 32), (1024, 48, 32), (256, 12, 64)):   # (64 x 64 cells: initial_resolution 512; only the row-streamed kernel covers it)
    grid = torch.nn.functional.normalize(torch.randn(1, T, H, H, 256, generator=g), dim=-1).to(dev)
    qf = torch.nn.functional.normalize(torch.randn(1, Q, 256, generator=g), dim=-1).to(dev)
    qp = torch.cat([torch.randint(0, T, (1, Q, 1), generator=g).float(), torch.rand(1, Q, 2, generator=g) * 8 * H], -1).to(dev)
    outs = {}
    for mode, name in (((1, 'workspace'), (2, 'fused_tiled'), (0, 'fused')) if H == 32 else ((0, 'fused'),)):
      assert lib.tapir_debug_set_cv_mode(ctx, mode) == 0
      pts = torch.empty(1, Q, T, 2, device=dev); occ = torch.empty(1, Q, T, device=dev); expd = torch.empty(1, Q, T, device=dev)

      def run(i):
        rc = lib.tapir_tracks_from_cost_volume(ctx, qf.data_ptr(), grid.data_ptr(), qp.data_ptr(), 1, Q, T, H, H,
                                               pts.data_ptr(), occ.data_ptr(), expd.data_ptr(), stream)
        assert rc == 0, lib.tapir_last_error(ctx)
      t = timeit(run, max(5, reps // 2), warm=2)
      outs[name] = (pts.clone(), occ.clone())
      row = dict(kernel=f'cost_volume_stage_{name}', Q=Q, T=T, cells=f'{H}x{H}', dtype=model.dtype, **t,
                 ns_per_map=round(t['med_us'] * 1e3 / (Q * T), 1))
      if name != 'workspace' and 'workspace' in outs:
        d = torch.linalg.norm(outs[name][0] - outs['workspace'][0], dim=-1)
        row['tracks_median_diff_px'] = float(d.median()); row['tracks_frac_within_0.05px'] = float((d < 0.05).float().mean())
        row['occ_max_diff'] = float((outs[name][1] - outs['workspace'][1]).abs().max())
      results.append(row)
      print(json.dumps(row), flush=True)
    assert lib.tapir_debug_set_cv_mode(ctx, 0) == 0


def bench_contraction(model, reps, results):
  """north_star's explicit kernel target -- the MFMA fraction of einsum('bnc,bthwc->tbnhw') (tapir_model.py:433) -- at
  the shapes where SURVEY.md 8d says it is meaningful: config 5's cost volume (M = 4096 queries, K = 256, N = 96 frames x
  32 x 32 cells = 98304: 206 GFLOP) and config 2 (M = 256, N = 49152: 6.4 GFLOP).  Two forms:
    fused    : the contraction phase of the row-streamed kernel alone (cost maps into LDS, nothing leaves the CU),
               forms 1 (8 maps per workgroup) and 2 (16 maps), TAPIR_CV_FORM;
    workspace: the tiled GEMM of the separate-launch path writing the f32 volume to HBM (gemm.hpp), chunks of <= 256 MiB."""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  stream = model._stream()
  g = torch.Generator(device='cpu').manual_seed(0)
  for Q, T in ((256, 48), (4096, 96)):
    grid = torch.nn.functional.normalize(torch.randn(1, T, 32, 32, 256, generator=g), dim=-1).to(dev)
    qf = torch.nn.functional.normalize(torch.randn(1, Q, 256, generator=g), dim=-1).to(dev)
    flops = 2.0 * Q * 256 * T * 1024
    scratch = torch.zeros(T * ((Q + 7) // 8) + 8, device=dev)

    def run(i):
      rc = lib.tapir_debug_contraction(ctx, qf.data_ptr(), grid.data_ptr(), 1, Q, T, 32, 32, scratch.data_ptr(), stream)
      assert rc == 0, lib.tapir_last_error(ctx)
    run(0)   # (casts the grid once)
    t = timeit(run, max(5, reps // 2), warm=2)
    form = int(os.environ.get('TAPIR_CV_FORM', '1'))
    row = dict(kernel='contraction_fused_phase', form=form, M=Q, K=256, N=T * 1024, dtype=model.dtype, **t,
               gflop=round(flops / 1e9, 1), tflops=round(flops / (t['med_us'] * 1e-6) / 1e12, 1),
               frac_of_bf16_mfma_peak=round(flops / (t['med_us'] * 1e-6) / 2.5e15, 4),
               note='includes the cast of the 256 query vectors and the zero fill of the cost maps; the 16-column B port '
                    'holds 8 (form 1) or 16 (forms 0, 2) queries')
    results.append(row)
    print(json.dumps(row), flush=True)
    # the separate-launch GEMM into a workspace (build_cost_volume), query chunks bounded by the 256-MiB volume
    qc = max(1, min(Q, (256 << 20) // (T * 1024 * 4)))
    vol = torch.empty(1, qc, T, 32, 32, device=dev)
    qsub = qf[:, :qc].contiguous()

    def run2(i):
      rc = lib.tapir_build_cost_volume(ctx, qsub.data_ptr(), grid.data_ptr(), 1, qc, T, 32, 32, 256, vol.data_ptr(), stream)
      assert rc == 0, lib.tapir_last_error(ctx)
    t2 = timeit(run2, max(5, reps // 2), warm=2)
    f2 = 2.0 * qc * 256 * T * 1024
    row = dict(kernel='contraction_gemm_to_workspace', M=qc, K=256, N=T * 1024, dtype=model.dtype, **t2,
               gflop=round(f2 / 1e9, 1), tflops=round(f2 / (t2['med_us'] * 1e-6) / 1e12, 1),
               frac_of_bf16_mfma_peak=round(f2 / (t2['med_us'] * 1e-6) / 2.5e15, 4),
               hbm_write_GBps=round(qc * T * 1024 * 4 / (t2['med_us'] * 1e-6) / 1e9, 1),
               note='writes the f32 volume: M*N*4 bytes; includes the operand casts')
    results.append(row)
    print(json.dumps(row), flush=True)


def trace_cv_fused(model):
  """per-phase shader-cycle totals of the row-streamed fused cost-volume kernel (costvol_rows.hpp), every wave of every
  workgroup; TRACE build (tapnet_amd/csrc/build.sh --exp, TAPIR_HIP_LIB=tools/bin/libtapir_hip_exp.so)"""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  Q, T = 256, 48
  g = torch.Generator(device='cpu').manual_seed(0)
  grid = torch.nn.functional.normalize(torch.randn(1, T, 32, 32, 256, generator=g), dim=-1).to(dev)
  qf = torch.nn.functional.normalize(torch.randn(1, Q, 256, generator=g), dim=-1).to(dev)
  pts = torch.empty(1, Q, T, 2, device=dev); occ = torch.empty(1, Q, T, device=dev); expd = torch.empty(1, Q, T, device=dev)
  form = int(os.environ.get('TAPIR_CV_FORM', '0'))
  nwg, nwv = T * (Q // (8 if form == 1 else 16)), (16 if form == 0 else 8)
  buf = torch.zeros(nwg * nwv * 8, dtype=torch.int64, device=dev)
  for it in range(2):
    assert lib.tapir_debug_set_trace(ctx, ctypes.c_void_p(buf.data_ptr())) == 0
    rc = lib.tapir_tracks_from_cost_volume(ctx, qf.data_ptr(), grid.data_ptr(), None, 1, Q, T, 32, 32,
                                           pts.data_ptr(), occ.data_ptr(), expd.data_ptr(), model._stream())
    assert rc == 0, lib.tapir_last_error(ctx)
    torch.cuda.synchronize()
  lib.tapir_debug_set_trace(ctx, None)
  t = buf.view(nwg * nwv, 8).double().cpu().numpy()
  names = {0: 'contraction + zero fill + constants', 1: 'barrier (cost maps complete)', 2: 'conv1 + relu + ring + conv2 (MFMA f32)',
           4: 'logit chain (DPP, bpermute, store)', 5: 'occlusion conv rows (MFMA)', 3: 'soft arg max + head tail'}
  tot = t.sum(-1).mean()
  print(f'row-streamed cost-volume phase trace ({model.dtype}): form {form}: mean over {nwg} workgroups x {nwv} waves ({2 if form == 2 else 1} map(s) per wave); total {tot:.0f} cycles per wave')
  for k in (0, 1, 2, 4, 5, 3):
    print(f'  {names[k]:42s} {t[:, k].mean():10.0f} cycles  {100 * t[:, k].mean() / tot:5.1f} %   waves min/max {t[:, k].min():.0f}/{t[:, k].max():.0f}')


def trace_fused(model):
  """per-phase shader-cycle totals of the fused mixer kernel (TRACE build: needs a library built with
  -DTAPIR_EXPERIMENTS, TAPIR_HIP_LIB=...): mean over waves of [in/out linear, LN1, token mixing,
  LN2 + xn, up + GELU + store, barrier, down, barrier] for N = 256 tracks x 48 frames."""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  N, T = 256, 48
  cin = 388 + 49 * (2 + model.pyramid_level)
  x = torch.randn(N, T, cin, device=dev)
  out = torch.empty(N, T, 388, device=dev)
  buf = torch.zeros(N * 8 * 8, dtype=torch.int64, device=dev)
  assert lib.tapir_debug_set_mixer_mode(ctx, 2) == 0
  for it in range(3):
    buf.zero_()
    assert lib.tapir_debug_set_trace(ctx, ctypes.c_void_p(buf.data_ptr())) == 0
    rc = lib.tapir_pips_mixer(ctx, x.data_ptr(), N, T, out.data_ptr(), None, None, None, None, model._stream())
    assert rc == 0, lib.tapir_last_error(ctx)
    torch.cuda.synchronize()
  lib.tapir_debug_set_trace(ctx, None)
  lib.tapir_debug_set_mixer_mode(ctx, 0)
  t = buf.view(N, 8, 8).double().cpu().numpy()
  names = ['in+out linear', 'LN1', 'token mixing', 'LN2 + xn write', 'up + GELU + h store', 'barrier (h visible)',
           'down', 'barrier (h free)']
  tot = t.sum(-1).mean()
  print(f'fused mixer phase trace ({model.dtype}): mean shader cycles per wave, whole kernel {tot:.0f} cycles')
  for k, nm in enumerate(names):
    per_wave = t[:, :, k].mean(0)
    print(f'  {nm:24s} {t[:, :, k].mean():10.0f} cycles  {100 * t[:, :, k].mean() / tot:5.1f} %   per block {t[:, :, k].mean() / 12:8.0f}'
          f'   waves min/max {per_wave.min():.0f}/{per_wave.max():.0f}')


def trace_wide(model):
  """per-phase shader-cycle totals of the WIDE fused mixer kernel (two tracks per workgroup; TRACE build:
  -DTAPIR_EXPERIMENTS library): [in/out linear, LN1 (+ parameter copy), token mixing, LN2 + xn, up, GELU + h store,
  down, barriers] for N = 1024 tracks x 48 frames."""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  N, T = 1024, 48
  cin = 388 + 49 * (2 + model.pyramid_level)
  x = torch.randn(N, T, cin, device=dev)
  out = torch.empty(N, T, 388, device=dev)
  nwg = N // 2
  buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
  assert lib.tapir_debug_set_mixer_mode(ctx, 3) == 0
  for it in range(3):
    buf.zero_()
    assert lib.tapir_debug_set_trace(ctx, ctypes.c_void_p(buf.data_ptr())) == 0
    rc = lib.tapir_pips_mixer(ctx, x.data_ptr(), N, T, out.data_ptr(), None, None, None, None, model._stream())
    assert rc == 0, lib.tapir_last_error(ctx)
    torch.cuda.synchronize()
  lib.tapir_debug_set_trace(ctx, None)
  lib.tapir_debug_set_mixer_mode(ctx, 0)
  t = buf.view(nwg, 8, 8).double().cpu().numpy()
  names = ['in+out linear', 'LN1 (+ parameter copy)', 'token mixing (2 tracks)', 'LN2 + xn write', 'up', 'GELU + h store',
           'down', 'barriers of the chunk loop']
  tot = t.sum(-1).mean()
  print(f'wide fused mixer phase trace ({model.dtype}), N = {N} x T = {T}: mean shader cycles per wave, whole kernel {tot:.0f} cycles')
  for k, nm in enumerate(names):
    per_wave = t[:, :, k].mean(0)
    print(f'  {nm:28s} {t[:, :, k].mean():10.0f} cycles  {100 * t[:, :, k].mean() / tot:5.1f} %   per block {t[:, :, k].mean() / 12:8.0f}'
          f'   waves min/max {per_wave.min():.0f}/{per_wave.max():.0f}')


def trace_conv(model):
  """per-phase shader-cycle totals of the fused 3x3 convolution (TRACE build: library built with
  -DTAPIR_EXPERIMENTS, TAPIR_HIP_LIB=...): mean over waves and workgroups."""
  lib, ctx = model._lib, model._ctx
  dev = model.device
  stream = model._stream()
  names = ['stage tile (norm+relu -> LDS)', 'barrier (tile visible)', 'k loop (9 taps)', 'barrier (tile free)',
           'epilogue (add, store, stats)', 'summary merge']
  for (n, h, w, c) in ((48, 128, 128, 64), (48, 64, 64, 128), (48, 32, 32, 256)):
    rows, tiles = ctypes.c_int(), ctypes.c_int()
    assert lib.tapir_conv_plan(ctx, h, w, c, c, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
    x = (torch.randn(n, h, w, c, device=dev) * 1.5 + 0.5).to(torch.bfloat16)
    sc = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    wh = (torch.randn(c, c, 3, 3) / (9 * c) ** 0.5).contiguous()
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.3
    part_in = torch.empty(n, 4, c, 2, device=dev)
    part_out = torch.empty(n, tiles.value, c, 2, device=dev)
    ss = torch.empty(n, c, 2, device=dev)
    y = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    ws = ctypes.c_void_p()
    assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wh.data_ptr()), c, c, 3, ctypes.byref(ws)) == 0
    assert lib.tapir_inorm_stats(ctx, x.data_ptr(), None, None, part_in.data_ptr(), n, h * w, c, 4, stream) == 0
    nwg = n * tiles.value
    buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
    for it in range(3):
      buf.zero_()
      assert lib.tapir_debug_set_trace(ctx, ctypes.c_void_p(buf.data_ptr())) == 0
      assert lib.tapir_conv_fused(ctx, x.data_ptr(), part_in.data_ptr(), 4, 0, gamma.data_ptr(), beta.data_ptr(),
                                  ss.data_ptr(), ws, sc.data_ptr(), y.data_ptr(), part_out.data_ptr(), n, h, w, c,
                                  c, 3, 1, stream) == 0
      torch.cuda.synchronize()
    lib.tapir_debug_set_trace(ctx, None)
    waves = 4 if rows.value * w <= 16384 // c else 8
    t = buf[:nwg * waves * 8].view(nwg, waves, 8).double().cpu().numpy()
    tot = t.sum(-1).mean()
    mf = 2.0 * rows.value * w * 9 * c * c
    print(f'conv3x3 fused phase trace, [{n},{h},{w},{c}] ({nwg} workgroups of {rows.value} rows, {waves} waves): mean shader cycles per '
          f'wave {tot:.0f}; MFMA floor {mf / 4069:.0f} cycles per workgroup')
    for k, nm in enumerate(names):
      pw = t[:, :, k].mean(0)
      print(f'  {nm:32s} {t[:, :, k].mean():9.0f} cycles {100 * t[:, :, k].mean() / tot:5.1f} %   waves min/max {pw.min():.0f}/{pw.max():.0f}')


def trace_conv_flat(model):
  """per-phase shader-cycle totals of the flat-tiled 256-channel convolution (conv_flat.hpp; -DTAPIR_EXPERIMENTS library)"""
  lib, ctx = model._lib, model._ctx
  dev, stream, c, h, w = model.device, model._stream(), 256, 32, 32
  names = ['stage tile (norm+relu -> LDS)', 'barrier (tile visible)', 'k loop (9 taps)', 'barrier (tile free)', 'epilogue (store, stats, merge)']
  assert lib.tapir_debug_set_conv_flat(ctx, 1) == 0
  for n in (6, 12, 48):
    rows, tiles = ctypes.c_int(), ctypes.c_int()
    assert lib.tapir_conv_plan(ctx, h, w, c, c, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
    x = (torch.randn(n, h, w, c, device=dev) * 1.5 + 0.5).to(torch.bfloat16)
    sc = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    wh = (torch.randn(c, c, 3, 3) / (9 * c) ** 0.5).contiguous()
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
    part_in = torch.empty(n, 4, c, 2, device=dev)
    part_out = torch.empty(n, tiles.value, c, 2, device=dev)
    ss = torch.empty(n, c, 2, device=dev)
    y = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    ws = ctypes.c_void_p()
    assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wh.data_ptr()), c, c, 3, ctypes.byref(ws)) == 0
    assert lib.tapir_inorm_stats(ctx, x.data_ptr(), None, None, part_in.data_ptr(), n, h * w, c, 4, stream) == 0
    nwg = -(-n * tiles.value // 3)
    buf = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
    for it in range(3):
      buf.zero_()
      assert lib.tapir_debug_set_trace(ctx, ctypes.c_void_p(buf.data_ptr())) == 0
      assert lib.tapir_conv_fused(ctx, x.data_ptr(), part_in.data_ptr(), 4, 0, gamma.data_ptr(), beta.data_ptr(), ss.data_ptr(), ws,
                                  sc.data_ptr(), y.data_ptr(), part_out.data_ptr(), n, h, w, c, c, 3, 1, stream) == 0
      torch.cuda.synchronize()
    lib.tapir_debug_set_trace(ctx, None)
    t = buf.view(nwg, 8, 8).double().cpu().numpy()
    tot = t.sum(-1).mean()
    print(f'conv3x3 FLAT phase trace, [{n},{h},{w},{c}] ({nwg} workgroups of 3 slabs = 192 pixels, 8 waves): mean shader cycles per wave '
          f'{tot:.0f}; MFMA floor {2.0 * 192 * 9 * c * c / 4069:.0f} cycles per workgroup')
    for k, nm in enumerate(names):
      pw = t[:, :, k].mean(0)
      print(f'  {nm:32s} {t[:, :, k].mean():9.0f} cycles {100 * t[:, :, k].mean() / tot:5.1f} %   waves min/max {pw.min():.0f}/{pw.max():.0f}')
    lib.tapir_conv_free(ctx, ws)
  assert lib.tapir_debug_set_conv_flat(ctx, -1) == 0


def bench_conv(model, reps, results):
  """the fused 3x3 backbone convolution (conv_fused.hpp) against the launches it replaces: finalize +
  normalise/ReLU kernel, the MIOpen convolution, the statistics (+ residual add) kernel"""
  import torch.nn.functional as F
  lib, ctx = model._lib, model._ctx
  dev = model.device
  stream = model._stream()
  torch.backends.cudnn.benchmark = True
  for (n, h, w, c) in ((48, 128, 128, 64), (48, 64, 64, 128), (48, 32, 32, 256), (8, 256, 256, 64)):
    rows, tiles = ctypes.c_int(), ctypes.c_int()
    assert lib.tapir_conv_plan(ctx, h, w, c, c, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
    x = [(torch.randn(n, h, w, c, device=dev) * 1.5 + 0.5).to(torch.bfloat16) for _ in range(3)]
    sc = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    wt = (torch.randn(c, c, 3, 3, device=dev) / (9 * c) ** 0.5)
    wcl = wt.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gamma = torch.rand(c, device=dev) + 0.5
    beta = torch.randn(c, device=dev) * 0.3
    slabs = max(1, min(h * w // 64, -(-1024 // n), 64))
    part_in = torch.empty(n, slabs, c, 2, device=dev)
    part_out = torch.empty(n, tiles.value, c, 2, device=dev)
    part2 = torch.empty(n, slabs, c, 2, device=dev)
    ss = torch.empty(n, c, 2, device=dev)
    y = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    yn = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    ws = ctypes.c_void_p()
    wh = wt.cpu().contiguous()
    assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wh.data_ptr()), c, c, 3, ctypes.byref(ws)) == 0
    for t_ in x:
      assert lib.tapir_inorm_stats(ctx, t_.data_ptr(), None, None, part_in.data_ptr(), n, h * w, c, slabs, stream) == 0
    def hip(i):
      assert lib.tapir_conv_fused(ctx, x[i % 3].data_ptr(), part_in.data_ptr(), slabs, 0, gamma.data_ptr(),
                                  beta.data_ptr(), ss.data_ptr(), ws, sc.data_ptr(), y.data_ptr(),
                                  part_out.data_ptr(), n, h, w, c, c, 3, 1, stream) == 0
    conv_out = [None]
    def miopen_conv(i):
      conv_out[0] = F.conv2d(yn.permute(0, 3, 1, 2), wcl, None, padding=1)
    def replaced(i):
      assert lib.tapir_inorm_relu(ctx, x[i % 3].data_ptr(), part_in.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                  ss.data_ptr(), yn.data_ptr(), None, n, h, w, c, slabs, 0, h, w, stream) == 0
      o = F.conv2d(yn.permute(0, 3, 1, 2), wcl, None, padding=1).permute(0, 2, 3, 1)
      assert lib.tapir_inorm_stats(ctx, o.data_ptr(), sc.data_ptr(), o.data_ptr(), part2.data_ptr(), n, h * w, c, slabs, stream) == 0
      conv_out[0] = o
    # correctness on the GPU: the fused kernel against the replaced sequence (same operand rounding;
    # the sequence rounds the convolution to bf16 before the add, the fused kernel only after it)
    assert lib.tapir_inorm_stats(ctx, x[0].data_ptr(), None, None, part_in.data_ptr(), n, h * w, c, slabs, stream) == 0
    hip(0); replaced(0)
    torch.cuda.synchronize()
    d = (y.float() - conv_out[0].float()).abs()
    flops = 2.0 * n * h * w * c * c * 9
    for name, fn in (('conv3x3_fused_hip', hip), ('miopen_conv_only', miopen_conv), ('norm_relu+miopen_conv+stats_add', replaced)):
      t = timeit(fn, reps)
      row = dict(kernel=name, shape=[n, h, w, c], tiles=tiles.value, rows=rows.value, **t,
                 tflops=round(flops / (t['med_us'] * 1e-6) / 1e12, 1),
                 max_abs_diff_vs_replaced=round(float(d.max()), 4), mean_abs_diff=round(float(d.mean()), 6))
      results.append(row)
      print(json.dumps(row), flush=True)


def bench_conv_flat(model, reps, results):
  """the flat tiling of the 3x3 256 -> 256 block convolutions (conv_flat.hpp) against the per-image tiling
  (conv_fused.hpp), alternated in one process: a frame group (12 frames), half a clip, the whole clip, and the dual
  conv_0 + proj_conv launch; bit-identity of the two outputs is checked on the way"""
  lib, ctx = model._lib, model._ctx
  dev, stream, c, h, w = model.device, model._stream(), 256, 32, 32
  for n in (6, 12, 24, 48, 96):
    rows, tiles = ctypes.c_int(), ctypes.c_int()
    assert lib.tapir_conv_plan(ctx, h, w, c, c, 3, 1, ctypes.byref(rows), ctypes.byref(tiles)) == 0
    x = [(torch.randn(n, h, w, c, device=dev) * 1.5 + 0.5).to(torch.bfloat16) for _ in range(3)]
    sc = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
    wt = (torch.randn(c, c, 3, 3) / (9 * c) ** 0.5).contiguous()
    w1 = (torch.randn(c, c, 1, 1) / c ** 0.5).contiguous()
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
    part_in = torch.empty(n, 4, c, 2, device=dev)
    part_out = torch.empty(n, tiles.value, c, 2, device=dev)
    ss, ssn = torch.empty(n, c, 2, device=dev), torch.empty(n, c, 2, device=dev)
    arrive = torch.zeros(n, dtype=torch.int32, device=dev)
    y = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    yp = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    ws, wsd = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.tapir_conv_pack(ctx, ctypes.c_void_p(wt.data_ptr()), c, c, 3, ctypes.byref(ws)) == 0
    assert lib.tapir_conv_pack_dual(ctx, ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(w1.data_ptr()), c, c, 1, ctypes.byref(wsd)) == 0
    assert lib.tapir_inorm_stats(ctx, x[0].data_ptr(), None, None, part_in.data_ptr(), n, h * w, c, 4, stream) == 0
    nn = _ffi.TapirNextNorm(gamma.data_ptr(), beta.data_ptr(), ssn.data_ptr(), arrive.data_ptr())
    def plain(i):
      assert lib.tapir_conv_fused_nn(ctx, x[i % 3].data_ptr(), None, 4, 0, gamma.data_ptr(), beta.data_ptr(), ss.data_ptr(), ws,
                                     sc.data_ptr(), y.data_ptr(), part_out.data_ptr(), n, h, w, c, c, 3, 1, ctypes.byref(nn), stream) == 0
    def dual(i):
      assert lib.tapir_conv_fused_dual_nn(ctx, x[i % 3].data_ptr(), None, 4, 0, gamma.data_ptr(), beta.data_ptr(), ss.data_ptr(), wsd,
                                          y.data_ptr(), yp.data_ptr(), part_out.data_ptr(), n, h, w, c, c, 1, ctypes.byref(nn), stream) == 0
    # (a, b) pairs once (part_in NULL afterwards: the launch alone is timed)
    assert lib.tapir_conv_fused_nn(ctx, x[0].data_ptr(), part_in.data_ptr(), 4, 0, gamma.data_ptr(), beta.data_ptr(), ss.data_ptr(), ws,
                                   sc.data_ptr(), y.data_ptr(), part_out.data_ptr(), n, h, w, c, c, 3, 1, ctypes.byref(nn), stream) == 0
    flops = 2.0 * n * h * w * c * c * 9
    for name, fn, fl in (('conv3x3_c256+shortcut', plain, flops), ('conv3x3_c256+proj (dual)', dual, flops * 10 / 9)):
      ref = None
      for rep in range(2):
        for mode, form in ((0, 'per-image tiles'), (1, 'flat')):
          assert lib.tapir_debug_set_conv_flat(ctx, mode) == 0
          fn(0); torch.cuda.synchronize()
          if ref is None:
            ref = (y.clone(), part_out.clone(), ssn.clone())
          same = bool(torch.equal(y, ref[0]) and torch.equal(part_out, ref[1]) and torch.equal(ssn, ref[2]))
          t = timeit(fn, reps)
          row = dict(kernel=name, form=form, shape=[n, h, w, c], rep=rep, **t, tflops=round(fl / (t['med_us'] * 1e-6) / 1e12, 1),
                     frac_of_bf16_peak=round(fl / (t['med_us'] * 1e-6) / 2.5e15, 3), bit_identical_to_first=same)
          results.append(row)
          print(json.dumps(row), flush=True)
    assert lib.tapir_debug_set_conv_flat(ctx, -1) == 0
    lib.tapir_conv_free(ctx, ws); lib.tapir_conv_free(ctx, wsd)


def bench_backbone(model, reps, results):
  if os.environ.get('TAPIR_CUDNN_BENCHMARK'):
    torch.backends.cudnn.benchmark = os.environ['TAPIR_CUDNN_BENCHMARK'] == '1'
  dev = model.device
  video = torch.as_tensor(synthetic.make_video(1, 48, 256, 256), device=dev)
  def run(i):
    model.get_feature_grids(video)
  t = timeit(run, max(3, reps // 8), warm=2)
  row = dict(kernel='backbone_get_feature_grids', dtype=model.dtype, **t)
  results.append(row)
  print(json.dumps(row), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--what', default='gemm,mix,mixer')
  ap.add_argument('--reps', type=int, default=40)
  ap.add_argument('--dtypes', default='bfloat16')
  ap.add_argument('--shapes', default='')
  ap.add_argument('--tiles', default='')
  ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'kbench.json'))
  args = ap.parse_args()
  what = set(args.what.split(','))
  results = []
  for dtype in args.dtypes.split(','):
    need_bb = 'backbone' in what
    w = synthetic.make_weights(0, 0, False, backbone=need_bb)
    model = tapir_model.TAPIR(pyramid_level=0, weights=w, dtype=dtype, device='cuda:0')
    if 'gemm' in what:
      bench_gemm(model, args.reps, results, set(args.shapes.split(',')) if args.shapes else None,
                 set(int(t) for t in args.tiles.split(',')) if args.tiles else None)
    if 'mix' in what:
      bench_mix(model, args.reps, results)
    if 'fill' in what:
      bench_fill(model, args.reps, results)
    if 'gemmtrace' in what:
      trace_gemm(model)
    if 'gemmsteps' in what:
      trace_gemm_steps(model)
    if 'cvtrace' in what:
      trace_cv(model)
    if 'norm' in what:
      bench_norm(model, args.reps, results)
    if 'cv' in what:
      bench_cv(model, args.reps, results)
    if 'convflattrace' in what:
      trace_conv_flat(model)
    if 'convflat' in what:
      bench_conv_flat(model, args.reps, results)
    if 'conv' in what:
      bench_conv(model, args.reps, results)
    if 'convtrace' in what:
      trace_conv(model)
    if 'cvfusedtrace' in what:
      trace_cv_fused(model)
    if 'contraction' in what:
      bench_contraction(model, args.reps, results)
    if 'fusedtrace' in what:
      trace_fused(model)
    if 'widetrace' in what:
      trace_wide(model)
    if 'mixtrace' in what:
      trace_mix(model)
    if 'mixer' in what:
      if os.environ.get('KBENCH_MIXER_SHAPES'):
        shapes = tuple(tuple(int(v) for v in t.split('x')) for t in os.environ['KBENCH_MIXER_SHAPES'].split(','))
        bench_mixer(model, args.reps, results, shapes)
      else:
        bench_mixer(model, args.reps, results)
    if need_bb:
      bench_backbone(model, args.reps, results)
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, 'w') as f:
    json.dump(results, f, indent=1)


if __name__ == '__main__':
  main()
