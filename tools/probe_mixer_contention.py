"""Few-row mixer (5 tracks x 9 frames) repeated under contention: (a) alone, (b) with a second stream of this process
running large matmuls, (c) with a second PROCESS on the same GPU doing the same.  Compares runs 1.. with run 1."""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def mixer_runs(m, x, n, gm):
  lib, ctx = m._lib, m._ctx
  assert lib.tapir_debug_set_gemm_mode(ctx, gm) == 0
  outs = []
  for _ in range(n):
    o = torch.empty(x.shape[0], x.shape[1], 388, device=x.device)
    assert lib.tapir_pips_mixer(ctx, x.data_ptr(), x.shape[0], x.shape[1], o.data_ptr(), None, None, None, None, m._stream()) == 0
    outs.append(o)
  torch.cuda.synchronize()
  return max(float((outs[1] - o).abs().max()) for o in outs[2:])


def body(tag, q=None):
  from tapnet_amd import synthetic, tapir_model
  dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
  w = synthetic.make_weights(17, pyramid_level=1, extra_convs=False, backbone=False)
  msgs = []
  for dt in ('bfloat16', 'float32'):
    m = tapir_model.TAPIR(pyramid_level=1, weights=w, device=dev, initial_resolution=(64, 64), dtype=dt)
    x = torch.randn(5, 9, 388 + 49 * 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    for gm in (1, 0):
      quiet = mixer_runs(m, x, 24, gm)
      side = torch.cuda.Stream(dev)
      a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
      with torch.cuda.stream(side):
        for _ in range(60):
          a @ a
      busy = mixer_runs(m, x, 24, gm)
      torch.cuda.synchronize()
      msgs.append(f'{tag} {dt} gemm mode {gm}: quiet {quiet:.3e} | beside a matmul stream of this process {busy:.3e}')
  if q is not None:
    q.put('\n'.join(msgs))
  return msgs


def worker(rank, q):
  try:
    body(f'two processes, rank {rank}:', q)
  except Exception:
    import traceback
    q.put(traceback.format_exc())


if __name__ == '__main__':
  print('\n'.join(body('one process:')))
  import torch.multiprocessing as mp
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  ps = [ctx.Process(target=worker, args=(r, q)) for r in range(2)]
  for p in ps: p.start()
  for _ in ps: print(q.get(timeout=600))
  for p in ps: p.join(60)
