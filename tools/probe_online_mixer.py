"""Phase stamps of the online model's persistent mixer launch (csrc/mixer_online.hpp): where a block's time goes.
Runs tapir_pips_mixer (256 points x 1 frame, causal, context in and out) with the trace buffer set and prints, per
segment, the mean over blocks and workgroups in microseconds (wall_clock64: 100 MHz)."""
import argparse
import ctypes
import sys, os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tapnet_amd import synthetic, tapir_model  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--points', type=int, default=256)
  ap.add_argument('--form', type=int, default=0, help='tapir_debug_set_gemm_mode 3 + 4 x form (bit 0: a weight slice per XCD instead of a cluster per XCD; bit 1: acquire + plain loads)')
  ap.add_argument('--reps', type=int, default=5)
  args = ap.parse_args()
  N, nb = args.points, 12
  w = synthetic.make_weights(3, 1, False, backbone=False)
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=False, weights=w, dtype='bfloat16', device='cuda:0', use_causal_conv=True)
  assert m._lib.tapir_debug_set_gemm_mode(m._ctx, 3 + 4 * args.form) == 0
  g = torch.Generator(device='cuda').manual_seed(1)
  x = torch.randn((N, 1, 388 + 49 * 3), device='cuda', generator=g)
  c1 = torch.randn((nb, N, 2, 512), device='cuda', generator=g)
  c2 = torch.randn((nb, N, 2, 2048), device='cuda', generator=g)
  o1, o2 = torch.zeros_like(c1), torch.zeros_like(c2)
  out = torch.empty((N, 1, 388), device='cuda')
  tr = torch.zeros((256, nb, 8), dtype=torch.int64, device='cuda')

  def run():
    m._check(m._lib.tapir_pips_mixer(m._ctx, x.data_ptr(), N, 1, out.data_ptr(), c1.data_ptr(), c2.data_ptr(), o1.data_ptr(),
                                     o2.data_ptr(), m._stream()), 'tapir_pips_mixer')
  for _ in range(3):
    run()
  torch.cuda.synchronize()
  seg = np.zeros((args.reps, 6))
  skew = np.zeros((args.reps, 2))
  for rep in range(args.reps):
    tr.zero_()
    assert m._lib.tapir_debug_set_trace(m._ctx, ctypes.c_void_p(tr.data_ptr())) == 0
    run()
    torch.cuda.synchronize()
    m._lib.tapir_debug_set_trace(m._ctx, None)
    t = tr.cpu().numpy().astype(np.float64) / 100.0          # us
    act = t[:, 0, 0] > 0                                      # workgroups that ran
    t = t[act]
    blk = t[:, 1:, :]                                        # (block 0 has the cold start)
    seg[rep, 0] = (blk[:, :, 1] - blk[:, :, 0]).mean()       # loads -> x in LDS
    seg[rep, 1] = (blk[:, :, 2] - blk[:, :, 1]).mean()       # row arithmetic, operand row stored
    seg[rep, 2] = (blk[:, :, 3] - blk[:, :, 2]).mean()       # barrier 1 (drain, arrive, wait for the slowest)
    seg[rep, 3] = (blk[:, :, 4] - blk[:, :, 3]).mean()       # MLP phase, slab stores issued
    seg[rep, 4] = (blk[:, :, 5] - blk[:, :, 4]).mean()       # barrier 2
    seg[rep, 5] = (blk[:, :, 5] - blk[:, :, 0]).mean()
    ncl = t.shape[0] // 32
    form_by_xcd = not (args.form & 1)      # (default placement: cluster = blockIdx % 8)
    ids = np.arange(256)[act]
    cl = ids % 8 if form_by_xcd else ids // 32
    s2 = [np.ptp(blk[cl == c, :, 2], axis=0).mean() for c in np.unique(cl)]
    s4 = [np.ptp(blk[cl == c, :, 4], axis=0).mean() for c in np.unique(cl)]
    skew[rep] = np.mean(s2), np.mean(s4)
    if rep == 0:
      xcc = tr.cpu().numpy()[:, 0, 6]
      print('XCC_ID of workgroups 0..31:', xcc[:32].tolist())
      print('workgroups per XCC:', np.bincount(xcc.astype(np.int64), minlength=8).tolist(),
            ' blockIdx % 8 == XCC_ID + const for all:', [int(np.all((ids % 8) == ((xcc[ids] + o) % 8))) for o in range(8)])
    total = t[:, -1, 5].max() - t[:, 0, 0].min()
    print(f'rep {rep}: launch {total:7.1f} us  block0 {np.mean(t[:, 0, 5] - t[:, 0, 0]):6.2f}', flush=True)
  names = ['loads -> x staged', 'row arithmetic + row stored', 'barrier 1', 'MLP phase (slab stores issued)', 'barrier 2', 'block']
  for n, v in zip(names, np.median(seg, axis=0)):
    print(f'{n:34s} {v:7.2f} us')
  print(f'arrival spread inside a cluster: at barrier 1 {np.median(skew[:, 0]):.2f} us, at barrier 2 {np.median(skew[:, 1]):.2f} us')
  word = ctypes.c_uint(0)
  assert m._lib.tapir_online_sync_error(m._ctx, ctypes.byref(word)) == 0 and word.value == 0


if __name__ == '__main__':
  main()
