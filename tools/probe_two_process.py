"""Root-cause hunt for the two-process bf16 nondeterminism of the few-row mixer (profiles/r05_two_process_nondeterminism.txt).

Two PROCESSES share the GPU; each runs the separate-launch mixer (5 tracks x 9 frames, bf16) N times on the same input,
beside a matmul stream of its own, and stops it behind launch group k = 1, 2, ... (tapir_debug_mixer_stop): after every run
the engine's workspaces are copied to the host and compared bit for bit with the first run's.  Reported per stop point:
the FIRST workspace that differs, where (rows / columns) and by how much.  Variants:
  poison   every workspace is filled with 0xFF bytes (a NaN pattern in f32 and bf16) and every CU's LDS with 0x7FC07FC0
           before each run -- garbage that is stable within one process becomes visible within one process;
  quiet    no matmul stream.
    python tools/probe_two_process.py [--runs 24] [--procs 2]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

NAMES = ['mlp_in', 'xa', 'xb', 'xn', 'hid', 'res', 'splitk']


def hip():
  h = ctypes.CDLL('libamdhip64.so')
  h.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
  h.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
  return h


def workspaces(lib, ctx):
  out = {}
  for i, n in enumerate(NAMES):
    p, b = ctypes.c_void_p(), ctypes.c_ulonglong()
    assert lib.tapir_debug_workspace(ctx, i, ctypes.byref(p), ctypes.byref(b)) == 0
    if p.value and b.value:
      out[n] = (p.value, b.value)
  return out


def snapshot(h, ws, used):
  snap = {}
  for n, (p, cap) in ws.items():
    nb = min(cap, used.get(n, cap))
    a = np.empty(nb, np.uint8)
    assert h.hipMemcpy(a.ctypes.data, p, nb, 2) == 0
    snap[n] = a
  return snap


def torch_control(tag, runs):
  """Control: plain PyTorch kernels (library GEMM + element-wise + LayerNorm on the mixer's shapes) repeated in the same
  setting, compared bit for bit with the first run."""
  dev = torch.device('cuda', 0)
  g = torch.Generator(device=dev).manual_seed(3)
  x = torch.randn(45, 512, device=dev, generator=g)
  w1 = torch.randn(2048, 512, device=dev, generator=g).to(torch.bfloat16)
  w2 = torch.randn(512, 2048, device=dev, generator=g).to(torch.bfloat16)
  a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
  side = torch.cuda.Stream(dev)
  def run():
    y = x
    for _ in range(12):
      h = torch.nn.functional.gelu(torch.nn.functional.layer_norm(y, (512,)).to(torch.bfloat16) @ w1.t(), approximate='tanh')
      y = y + (h @ w2.t()).float() * 1e-2
    return y
  first, bad = None, 0
  for r in range(runs):
    with torch.cuda.stream(side):
      for _ in range(6):
        a @ a
    y = run()
    torch.cuda.synchronize()
    if first is None:
      first = y.clone()
    elif not torch.equal(y, first):
      bad += 1
  return f'{tag} torch control (12 x LayerNorm + bf16 GEMM + GELU + bf16 GEMM on [45, 512], beside a matmul stream): {bad}/{runs - 1} repeats differ'


def body(tag, runs, dtype='bfloat16', stops=(1, 2, 3, 4, 7, 0), variants=('busy', 'busy+poison', 'quiet'), modes=(1, 0), dump=None):
  from tapnet_amd import synthetic, tapir_model
  dev = torch.device('cuda', 0)
  torch.cuda.set_device(dev)
  h = hip()
  w = synthetic.make_weights(17, pyramid_level=1, extra_convs=False, backbone=False)
  m = tapir_model.TAPIR(pyramid_level=1, weights=w, device=dev, initial_resolution=(64, 64), dtype=dtype)
  lib, ctx = m._lib, m._ctx
  N, T = 5, 9
  R = N * T
  es = 2 if dtype == 'bfloat16' else 4
  k0 = 640 if es == 2 else 576
  used = dict(mlp_in=R * k0 * es, xa=R * 512 * 4, xb=R * 512 * 4, xn=R * 512 * es, hid=R * 2048 * es, res=R * 388 * 4)
  x = torch.randn(N, T, 388 + 49 * 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
  o = torch.empty(N, T, 388, device=dev)
  call = lambda: lib.tapir_pips_mixer(ctx, x.data_ptr(), N, T, o.data_ptr(), None, None, None, None, m._stream())
  assert call() == 0
  torch.cuda.synchronize()
  ws = workspaces(lib, ctx)
  a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
  side = torch.cuda.Stream(dev)
  msgs = []
  for gm in modes:
    assert lib.tapir_debug_set_gemm_mode(ctx, gm) == 0
    for variant in variants:
      for stop in stops:
        assert lib.tapir_debug_mixer_stop(ctx, stop) == 0
        first, bad = None, []
        for r in range(runs):
          if 'poison' in variant:
            for n, (p, cap) in ws.items():
              assert h.hipMemset(p, 0xFF, cap) == 0
            assert lib.tapir_debug_poison_lds(ctx, 0x7FC07FC0, m._stream()) == 0
          if 'busy' in variant:
            with torch.cuda.stream(side):
              for _ in range(6):
                a @ a
          assert call() == 0
          torch.cuda.synchronize()
          snap = snapshot(h, ws, used)
          if stop == 0:
            snap['out'] = o.cpu().numpy().view(np.uint8).ravel().copy()
          if first is None:
            first = snap
            if 'poison' in variant:   # anything of the USED part still poisoned after the run was never written: fine, but
              pass                    # a NaN in the output would be a poisoned read
            continue
          for n in snap:
            if n == 'splitk' and gm == 1:
              continue
            if not np.array_equal(snap[n], first[n]):
              idx = np.nonzero(snap[n] != first[n])[0]
              bad.append((r, n, idx))
              if len(bad) == 1:
                bad_snap = snap
              break
        nan_out = bool(np.isnan(o.cpu().numpy()).any()) if stop == 0 else None
        if bad:
          r, n, idx = bad[0]
          wbytes = dict(mlp_in=k0 * es, xa=2048, xb=2048, xn=512 * es, hid=2048 * es, res=388 * 4, out=388 * 4, splitk=4).get(n, 4)
          rows = sorted(set((idx // wbytes).tolist()))
          cols = sorted(set(((idx % wbytes) // (es if n in ('mlp_in', 'xn', 'hid') else 4)).tolist()))
          is16 = n in ('mlp_in', 'xn', 'hid') and es == 2
          def as_f32(b):
            if is16:
              return (b.view(np.uint16).astype(np.uint32) << 16).view(np.float32)
            return b.view(np.float32)
          va, vb = as_f32(first[n]), as_f32(bad_snap[n])
          e = np.nonzero(va.view(np.uint32) != vb.view(np.uint32))[0]
          samples = ', '.join(f'[{int(i)}] {va[i]:.6g} -> {vb[i]:.6g}' for i in e[:4])
          mag = f'max |diff| {float(np.nanmax(np.abs(va[e] - vb[e]))):.3e}; {samples}'
          detail = f'rows {rows[:12]}{"..." if len(rows) > 12 else ""} ({len(rows)}), cols {cols[:12]}{"..." if len(cols) > 12 else ""} ({len(cols)})'
          msgs.append(f'{tag} {dtype} gemm_mode {gm} {variant} stop {stop}: {len(bad)}/{runs - 1} runs differ; first: run {r}, '
                      f'workspace {n}, {len(idx)} bytes, {detail}; {mag}')
          if dump is not None and stop in (1, 2, 3, 4):
            fn = os.path.join(dump, f'{tag.strip("[] ").replace(" ", "_").replace(",", "")}_gm{gm}_{variant.replace("+", "_")}_stop{stop}.npz')
            np.savez_compressed(fn, x=x.cpu().numpy(), **{'good_' + k: v for k, v in first.items() if k != 'splitk'},
                                **{'bad_' + k: v for k, v in bad_snap.items() if k != 'splitk'})
            msgs.append(f'    dumped {fn}')
        else:
          msgs.append(f'{tag} {dtype} gemm_mode {gm} {variant} stop {stop}: all {runs - 1} repeats bit-identical'
                      + (f' (NaN in output: {nan_out})' if nan_out is not None else ''))
  assert lib.tapir_debug_mixer_stop(ctx, 0) == 0
  return msgs


def worker(rank, runs, q, kw):
  try:
    msgs = body(f'[2 procs, rank {rank}]', runs, **kw)
    msgs.append(torch_control(f'[2 procs, rank {rank}]', 200))
    q.put('\n'.join(msgs))
  except Exception:
    import traceback
    q.put(traceback.format_exc())


if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--runs', type=int, default=24)
  ap.add_argument('--procs', type=int, default=2)
  ap.add_argument('--skip-single', action='store_true')
  ap.add_argument('--stops', default='1,2,3,4,7,0')
  ap.add_argument('--variants', default='busy,busy+poison,quiet')
  ap.add_argument('--modes', default='1,0')
  ap.add_argument('--dump', default=None, help='directory for the snapshots of the first differing run per stop point (npz)')
  args = ap.parse_args()
  kw = dict(stops=tuple(int(v) for v in args.stops.split(',')), variants=tuple(args.variants.split(',')),
            modes=tuple(int(v) for v in args.modes.split(',')), dump=args.dump)
  if args.dump:
    os.makedirs(args.dump, exist_ok=True)
  if not args.skip_single:
    print('\n'.join(body('[1 proc]', args.runs, **kw)), flush=True)
    print(torch_control('[1 proc]', 200), flush=True)
  import torch.multiprocessing as mp
  mpc = mp.get_context('spawn')
  q = mpc.Queue()
  ps = [mpc.Process(target=worker, args=(r, args.runs, q, kw)) for r in range(args.procs)]
  for p in ps:
    p.start()
  for _ in ps:
    print(q.get(timeout=1500), flush=True)
  for p in ps:
    p.join(60)
