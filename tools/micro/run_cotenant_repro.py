"""Stand-alone reproducer of the co-tenant fault (profiles/r06_cotenant_fault.txt): nothing of the engine is loaded.

VICTIM   tools/micro/pk_forms_victim.hip: packed-f32 VALU forms, each checked against exactly-rounded scalar arithmetic.
AGGRESSOR tools/micro/cotenant_aggressors.hip: one small kernel launched again and again (kind 3 = back-to-back
         v_mfma_f32_16x16x32_bf16, C/D in AGPRs; 4 = the same with C/D in VGPRs; 0 = packed / scalar VALU only).
Settings:  alone    the victim by itself
           streams  ONE process: the victim on one stream, the aggressor on another
           procs    TWO processes on the one GPU: rank 0 the victim, rank 1 the aggressor
    python tools/micro/run_cotenant_repro.py [--kinds 0,3,4] [--launches 300]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
FORMS = ['pk_fma (no modifiers)', 'pk_fma op_sel_hi:[1,0,1]', 'pk_fma op_sel:[0,1,0]', 'pk_fma op_sel_hi:[1,0,1] -> dependent pk_fma op_sel:[0,1,0]',
         'pk_mul', 'pk_add', 'pk_fma op_sel:[0,1,0] accumulating in place', 'scalar v_fma_f32 x 2 (control)']
KINDS = ['valu only', 'AGPRs allocated, untouched', 'v_accvgpr traffic', 'MFMA, C/D in AGPRs', 'MFMA, C/D in VGPRs', 'ds on AGPRs', 'gemm_small mimic']


def load():
  v = ctypes.CDLL(os.path.join(HERE, 'libpkforms.so'))
  v.pk_forms_check.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
  a = ctypes.CDLL(os.path.join(HERE, 'libcotenant.so'))
  a.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
  return v, a


def victim(tag, launches, blocks, iters, aggr_kind=None, seconds=3.0):
  """aggr_kind != None: the aggressor runs in THIS process on a second stream."""
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(dev)
  v, a = load()
  seed = (torch.randn(4096, device=dev, generator=torch.Generator(device=dev).manual_seed(5)) * 0.7).contiguous()
  rep = torch.zeros(20, dtype=torch.int64, device=dev)
  buf = (torch.randn(1 << 20, device=dev) * 0.1).contiguous()
  side = torch.cuda.Stream(dev)
  s0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  s1 = ctypes.c_void_p(side.cuda_stream)
  t0 = time.perf_counter()
  i = 0
  while i < launches or time.perf_counter() - t0 < seconds:   # at least `launches`, and for at least `seconds`
    i += 1
    if aggr_kind is not None:
      for _ in range(8):
        assert a.aggr_launch(aggr_kind, 32, 400, buf.data_ptr(), s1) == 0
    assert v.pk_forms_check(seed.data_ptr(), blocks, iters, rep.data_ptr(), s0) == 0
    if i % 16 == 15:
      torch.cuda.synchronize()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  r = rep.cpu().numpy()
  bad = r[:16].reshape(8, 2)
  launches = i
  ops = launches * blocks * 256 * iters
  lines = [f'{tag}: {launches} launches x {blocks} blocks x 256 threads x {iters} trips = {ops:.2e} executions of every form ({dt:.1f} s)']
  for f in range(8):
    lines.append(f'    {FORMS[f]:72s} wrong low halves {int(bad[f, 0]):>10d}   wrong high halves {int(bad[f, 1]):>10d}')
  lines.append(f'    faulty results per quarter wave (lanes 0-15, 16-31, 32-47, 48-63): {r[16:20].tolist()}')
  return '\n'.join(lines)


def aggressor(tag, kind, seconds):
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(dev)
  _, a = load()
  buf = (torch.randn(1 << 20, device=dev) * 0.1).contiguous()
  s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  t0, n = time.perf_counter(), 0
  while time.perf_counter() - t0 < seconds:
    for _ in range(200):
      assert a.aggr_launch(kind, 32, 400, buf.data_ptr(), s) == 0
      n += 1
    torch.cuda.synchronize()
  return f'{tag}: aggressor kind {kind} ({KINDS[kind]}), 32 x 256 threads x 400 trips, {n} launches in {seconds:.0f} s'


def worker(fn, kw, q):
  try:
    q.put(fn(**kw))
  except Exception:
    import traceback
    q.put(traceback.format_exc())


if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--kinds', default='0,3,4')
  ap.add_argument('--launches', type=int, default=300)
  ap.add_argument('--blocks', type=int, default=64)
  ap.add_argument('--iters', type=int, default=2000)
  ap.add_argument('--settings', default='alone,streams,procs')
  args = ap.parse_args()
  base = dict(launches=args.launches, blocks=args.blocks, iters=args.iters)
  kinds = [int(k) for k in args.kinds.split(',')]
  sets = args.settings.split(',')
  if 'alone' in sets:
    print(victim('[alone]', **base), flush=True)
  if 'streams' in sets:
    for k in kinds:
      print(victim(f'[one process, aggressor kind {k} ({KINDS[k]}) on a second stream]', aggr_kind=k, **base), flush=True)
  if 'procs' in sets:
    import torch.multiprocessing as mp
    mpc = mp.get_context('spawn')
    for k in kinds:
      q = mpc.Queue()
      ps = [mpc.Process(target=worker, args=(victim, dict(tag=f'[two processes, rank 0 victim beside aggressor kind {k} ({KINDS[k]})]', seconds=18.0, **base), q)),
            mpc.Process(target=worker, args=(aggressor, dict(tag='[two processes, rank 1]', kind=k, seconds=25.0), q))]
      for p in ps:
        p.start()
      for _ in ps:
        print(q.get(timeout=900), flush=True)
      for p in ps:
        p.join(60)
