"""The fault behind the two-process nondeterminism of the few-row mixer, isolated to ONE kernel: mix_kernel alone
(tapir_debug_mix, block 0, T = 9 frames -> the general token-mixing kernel), launched again and again on the same
input and compared bit for bit with its first result.  tools/probe_two_process.py (profiles/r06_probe_two_process*.txt)
traced every differing run of the whole mixer to this kernel's x_out: 16 even channels of one wave's lanes 48..63 over
1-3 consecutive frames of one track.  This probe measures WHEN that happens and what the faulty values look like:

  --setting quiet       one process, nothing else on the GPU
  --setting stream      one process, a bf16 matmul stream of its own beside the probe
  --setting procs       two processes, each probing beside its own matmul stream (the original situation)
  --setting procs-quiet two processes, both probing, no matmul streams
  --setting neighbour   two processes: rank 0 probes with NO stream of its own, rank 1 only runs matmuls

Per setting: launches, faulty launches, and per fault the (track, frames, wave, lanes, channel parity) signature.
TAPIR_HIP_LIB selects another build of the library (compiler-flag variants of the same sources).
    python tools/probe_mix_fault.py --setting quiet,stream,procs --launches 3000 --tracks 64
"""
import argparse
import collections
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def probe(tag, launches, tracks, frames, busy, probe_here=True, seconds=None, dtype='bfloat16', stop=0, gemm_mode=1, matmuls=True, lib_path=None):
  if lib_path:   # (this rank's build of the library; before tapnet_amd is imported)
    os.environ['TAPIR_HIP_LIB'] = os.path.abspath(lib_path)
    tag = f'{tag} [{os.path.basename(lib_path)}]'
  from tapnet_amd import synthetic, tapir_model
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(dev)
  w = synthetic.make_weights(17, pyramid_level=1, extra_convs=False, backbone=False)
  m = tapir_model.TAPIR(pyramid_level=1, weights=w, device=dev, initial_resolution=(64, 64), dtype=dtype)
  lib, ctx = m._lib, m._ctx
  N, T = tracks, frames
  x = torch.randn(N, T, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
  xb = torch.empty(N, T, 512, device=dev)
  xn = torch.empty(N, T, 512, device=dev, dtype=torch.bfloat16 if dtype == 'bfloat16' else torch.float32)
  a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
  side = torch.cuda.Stream(dev)
  s = m._stream()
  call = lambda: lib.tapir_debug_mix(ctx, 0, x.data_ptr(), xb.data_ptr(), xn.data_ptr(), N, T, 0, s)
  if stop:
    # the kernel in its context: the separate-launch mixer up to launch group `stop` (1 input Linear, 2 + token mixing of
    # block 0, 3 + up-projection, 4 + down-projection), x_out / LN2(x_out) copied out of the engine's workspaces
    h = ctypes.CDLL('libamdhip64.so')
    h.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    xin = torch.randn(N, T, 388 + 49 * 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    o = torch.empty(N, T, 388, device=dev)
    assert lib.tapir_debug_set_gemm_mode(ctx, gemm_mode) == 0 and lib.tapir_debug_mixer_stop(ctx, stop) == 0
    assert lib.tapir_pips_mixer(ctx, xin.data_ptr(), N, T, o.data_ptr(), None, None, None, None, s) == 0
    torch.cuda.synchronize()
    ws = {}
    for i, n in ((2, 'xb'), (3, 'xn')):
      pp, bb = ctypes.c_void_p(), ctypes.c_ulonglong()
      assert lib.tapir_debug_workspace(ctx, i, ctypes.byref(pp), ctypes.byref(bb)) == 0 and pp.value
      ws[n] = pp.value
    def call():
      rc = lib.tapir_pips_mixer(ctx, xin.data_ptr(), N, T, o.data_ptr(), None, None, None, None, s)
      assert h.hipMemcpyAsync(xb.data_ptr(), ws['xb'], xb.numel() * 4, 3, s) == 0
      assert h.hipMemcpyAsync(xn.data_ptr(), ws['xn'], xn.numel() * xn.element_size(), 3, s) == 0
      return rc
  if not probe_here:   # the neighbour: for as long as the other rank probes, matmuls and / or its own launches (never checked)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
      for _ in range(50):
        if matmuls:
          with torch.cuda.stream(side):
            a @ a
        if busy:
          assert call() == 0
          n += 1
      torch.cuda.synchronize()
    what = (f'mixer up to launch group {stop} (gemm mode {gemm_mode})' if stop else 'mix_kernel') if busy else 'nothing of the engine'
    return f'{tag} for {seconds:.0f} s: {what} x {n}{", matmuls" if matmuls else ""}'
  assert call() == 0
  torch.cuda.synchronize()
  ref, ref_n = xb.clone(), xn.clone()
  faults = []
  t0 = time.perf_counter()
  for r in range(launches):
    if busy:
      with torch.cuda.stream(side):
        for _ in range(6):
          a @ a
    xb.fill_(float('nan'))
    assert call() == 0
    same = torch.equal(xb, ref) and torch.equal(xn, ref_n)
    if not same:
      d = (xb.view(torch.int32) != ref.view(torch.int32)).cpu().numpy()
      tr, fr, ch = np.nonzero(d)
      err = (xb - ref).abs().max().item()
      faults.append(dict(launch=r, tracks=sorted(set(tr.tolist())), frames=sorted(set(fr.tolist())), channels=sorted(set(ch.tolist())),
                         max_abs=err, xn_only=len(tr) == 0))
  dt = time.perf_counter() - t0
  torch.cuda.synchronize()
  what = f'mixer up to launch group {stop}' if stop else 'mix_kernel'
  lines = [f'{tag} {"beside a matmul stream" if busy else "no stream of its own"}: {launches} launches of {what} x {N} tracks '
           f'({dt:.1f} s): {len(faults)} faulty launches']
  sig = collections.Counter()
  for f in faults:
    ch = f['channels']
    waves = sorted(set(c // 128 for c in ch))
    lanes = sorted(set((c // 2) % 64 for c in ch))
    parity = sorted(set(c % 2 for c in ch))
    sig[(tuple(waves), (lanes[0], lanes[-1]) if lanes else (), tuple(parity), len(f['frames']), len(f['tracks']))] += 1
  for k, v in sorted(sig.items(), key=lambda kv: -kv[1]):
    lines.append(f'    {v} x  waves {list(k[0])}, lanes {k[1][0] if k[1] else "-"}..{k[1][1] if k[1] else "-"}, channel parity {list(k[2])}, '
                 f'{k[3]} frame(s), {k[4]} track(s)')
  for f in faults[:6]:
    lines.append(f'    launch {f["launch"]}: track {f["tracks"]}, frames {f["frames"]}, channels {f["channels"][:4]}..{f["channels"][-1:] } '
                 f'({len(f["channels"])}), max |diff| {f["max_abs"]:.3g}')
  return '\n'.join(lines)


def micro_aggressor(tag, kind, seconds, grid=32, iters=400):
  """rank 1 of --setting pair with --aggr micro:<kind>[:grid[:iters]]: one kernel of tools/micro/cotenant_aggressors.hip launched
  again and again (nothing of the engine is loaded in this process)."""
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(dev)
  lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'libcotenant.so'))
  lib.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
  buf = (torch.randn(1 << 20, device=dev) * 0.1).contiguous()   # 4 MiB
  s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
  names = ['valu (no AGPRs)', 'alloc (16 AGPRs allocated, untouched)', 'accmov (v_accvgpr_write / mov / read)', 'mfma_a (MFMA C/D in AGPRs)',
           'mfma_v (MFMA C/D in VGPRs)', 'ds_a (ds_write_b128 / ds_read_b128 on AGPRs)', 'mimic (loads -> MFMA in AGPRs -> ds_write from AGPRs -> pk_add)']
  t0, n = time.perf_counter(), 0
  while time.perf_counter() - t0 < seconds:
    for _ in range(200):
      assert lib.aggr_launch(kind, grid, iters, buf.data_ptr(), s) == 0
      n += 1
    torch.cuda.synchronize()
  return f'{tag} for {seconds:.0f} s: micro kernel {kind} = {names[kind]}, grid {grid} x 256 threads, {iters} trips, x {n}'


def worker(rank, kw, q):
  try:
    q.put(micro_aggressor(**kw['micro']) if 'micro' in kw else probe(**kw))
  except Exception:
    import traceback
    q.put(traceback.format_exc())


def two(args, kws):
  import torch.multiprocessing as mp
  mpc = mp.get_context('spawn')
  q = mpc.Queue()
  ps = [mpc.Process(target=worker, args=(r, kw, q)) for r, kw in enumerate(kws)]
  for p in ps:
    p.start()
  for _ in ps:
    print(q.get(timeout=3000), flush=True)
  for p in ps:
    p.join(60)


if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--setting', default='quiet,stream,procs')
  ap.add_argument('--launches', type=int, default=3000)
  ap.add_argument('--tracks', type=int, default=64)
  ap.add_argument('--frames', type=int, default=9)
  ap.add_argument('--dtype', default='bfloat16')
  ap.add_argument('--stop', type=int, default=0, help='0: mix_kernel alone; k: the separate-launch mixer up to launch group k')
  ap.add_argument('--aggr', default='4,1,1', help="setting 'pair': what rank 1 runs beside the probing rank 0: stop,gemm_mode,matmuls (stop -1: no engine launches)")
  ap.add_argument('--victim-matmuls', type=int, default=1)
  ap.add_argument('--victim-lib', default=None, help="setting 'pair': rank 0's build of the library")
  ap.add_argument('--aggr-lib', default=None, help="setting 'pair': rank 1's build of the library")
  args = ap.parse_args()
  base = dict(launches=args.launches, tracks=args.tracks, frames=args.frames, dtype=args.dtype, stop=args.stop)
  print(f'library: {os.environ.get("TAPIR_HIP_LIB", "tapnet_amd/csrc/libtapir_hip.so")}', flush=True)
  for st in args.setting.split(','):
    if st == 'quiet':
      print(probe('[1 proc]', busy=False, **base), flush=True)
    elif st == 'stream':
      print(probe('[1 proc]', busy=True, **base), flush=True)
    elif st == 'procs':
      two(args, [dict(tag=f'[2 procs, rank {r}]', busy=True, **base) for r in range(2)])
    elif st == 'procs-quiet':
      two(args, [dict(tag=f'[2 procs, rank {r}]', busy=False, **base) for r in range(2)])
    elif st == 'neighbour':
      secs = max(20.0, args.launches * 0.004)
      two(args, [dict(tag='[2 procs, rank 0 probes]', busy=False, **base),
                 dict(tag='[2 procs, rank 1 neighbour]', busy=True, probe_here=False, seconds=secs, **base)])
    elif st == 'pair':   # rank 0 probes (--stop), rank 1 runs --aggr unchecked
      secs = max(25.0, args.launches * 0.0035)
      if args.aggr.startswith('micro:'):
        f = [int(v) for v in args.aggr.split(':')[1:]]
        two(args, [dict(tag=f'[pair, rank 0 probes, aggressor {args.aggr}]', busy=bool(args.victim_matmuls), lib_path=args.victim_lib, **base),
                   dict(micro=dict(tag='[pair, rank 1 aggressor]', kind=f[0], seconds=secs, grid=f[1] if len(f) > 1 else 32,
                                   iters=f[2] if len(f) > 2 else 400))])
        continue
      a_stop, a_gemm, a_mm = (int(v) for v in args.aggr.split(','))
      two(args, [dict(tag=f'[pair, rank 0 probes, aggressor {args.aggr}]', busy=bool(args.victim_matmuls), lib_path=args.victim_lib, **base),
                 dict(tag='[pair, rank 1 aggressor]', busy=a_stop >= 0, probe_here=False, seconds=secs, matmuls=bool(a_mm), lib_path=args.aggr_lib,
                      **dict(base, stop=max(a_stop, 0), gemm_mode=a_gemm))])
    else:
      raise SystemExit(f'unknown setting {st}')
