"""The online tracker (persistent mixer launch, csrc/mixer_online.hpp) next to other work on the same GPU: a second stream of the
process (and, with --procs 2, a second process) keeps launching MFMA kernels of `grid` workgroups while the session runs.  The
persistent launch needs its 256 workgroups on the device at once; co-tenant kernels that END let it in sooner or later (its waits are
bounded at ~2 s), so every frame must come out finite and bit-identical to the quiet run -- or the error word must say so."""
import argparse
import ctypes
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tapnet_amd import online, synthetic, tapir_model  # noqa: E402


def aggressors():
  lib = ctypes.CDLL(os.path.join(ROOT, 'tools', 'micro', 'libcotenant.so'))
  lib.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
  return lib


def aggressor_process(seconds, grid, iters):
  lib = aggressors()
  buf = (torch.randn(1 << 20, device='cuda') * 0.1).contiguous()
  s = torch.cuda.Stream()
  t0 = time.time()
  while time.time() - t0 < seconds:
    for _ in range(16):
      lib.aggr_launch(3, grid, iters, buf.data_ptr(), ctypes.c_void_p(s.cuda_stream))
    s.synchronize()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--frames', type=int, default=400)
  ap.add_argument('--grid', type=int, default=512)
  ap.add_argument('--iters', type=int, default=4000)
  ap.add_argument('--procs', type=int, default=1)
  ap.add_argument('--aggressor', type=float, default=0.0)
  args = ap.parse_args()
  if args.aggressor > 0:
    return aggressor_process(args.aggressor, args.grid, args.iters)
  S, Q = 256, 256
  w = synthetic.make_weights(0, pyramid_level=1, extra_convs=True)
  m = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, use_causal_conv=True, weights=w, dtype='bfloat16', device='cuda:0')
  video = torch.as_tensor(synthetic.make_video(1, 8, S, S)).cuda()
  qp = torch.as_tensor(synthetic.make_queries(2, Q, 1, S, S)).cuda()
  lib = aggressors()
  buf = (torch.randn(1 << 20, device='cuda') * 0.1).contiguous()
  side = torch.cuda.Stream()

  def session(use_graph, noisy):
    trk = online.OnlineTracker(m, Q, (S, S), use_graph=use_graph)
    trk.init(video[:, :1], qp)
    outs = []
    t0 = time.perf_counter()
    for t in range(args.frames):
      if noisy and t % 2 == 0:
        for _ in range(4):
          lib.aggr_launch(3, args.grid, args.iters, buf.data_ptr(), ctypes.c_void_p(side.cuda_stream))
      outs.append(trk.step(video[:, t % 8:t % 8 + 1])['tracks'].clone())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.frames * 1e3
    err = None
    try:
      trk.check()
    except RuntimeError as e:
      err = str(e)
    trk.close()
    return outs, ms, err
  ref, ms0, err0 = session(True, False)
  print(f'quiet, replay: {ms0:.3f} ms per frame, error word: {err0}', flush=True)
  procs = []
  if args.procs > 1:
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--aggressor', '60', '--grid', str(args.grid),
                               '--iters', str(args.iters)]) for _ in range(args.procs - 1)]
    time.sleep(8)
  for use_graph in (True, False):
    outs, ms, err = session(use_graph, True)
    finite = sum(bool(torch.isfinite(o).all()) for o in outs)
    same = sum(bool(torch.equal(a, b)) for a, b in zip(outs, ref))
    print(f'next to MFMA kernels of {args.grid} workgroups on a second stream' + (f' and {args.procs - 1} other process(es)' if procs else '') +
          f', {"replay" if use_graph else "eager"}: {ms:.3f} ms per frame, {finite} / {len(outs)} frames finite, {same} / {len(outs)} '
          f'bit-identical to the quiet run, error word: {err}', flush=True)
  for p in procs:
    p.kill()


if __name__ == '__main__':
  main()
