"""`-m gpu`: the multi-GPU path executed with the REAL model: two processes run
tapnet_amd.distributed.sharded_call (frame-sharded HIP/MIOpen backbone -> all-gather of the
feature grids along T -> query-sharded HIP hot path -> gather of the outputs).

With >= 2 visible devices the ranks take one GPU each and the collectives are RCCL ("nccl");
on a one-GPU box both ranks share the device ("run ranks as processes on the
available devices for correctness") and, because RCCL refuses two ranks on one device, the
collectives go through gloo with host staging -- the same code path otherwise.

Checks per rank:
  * sharded result == the unsharded call on the SAME gathered grids, bitwise (queries are
    independent units and the hot path has no batch-dependent arithmetic);
  * sharded result ~= a plain single-process call (not bitwise: MIOpen convolutions accumulate
    with atomics, two backbone runs differ in the last bits).
"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, T, Q, q):
  try:
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch.distributed as dist
    from tapnet_amd import distributed as tdist, synthetic, tapir_model
    ndev = torch.cuda.device_count()
    dev = torch.device('cuda', rank % ndev)
    torch.cuda.set_device(dev)
    backend = 'nccl' if ndev >= world else 'gloo'
    dist.init_process_group(backend, rank=rank, world_size=world)
    S = 64
    w = synthetic.make_weights(17, pyramid_level=1, extra_convs=False)
    m = tapir_model.TAPIR(pyramid_level=1, weights=w, device=dev, initial_resolution=(S, S))
    video = torch.as_tensor(synthetic.make_video(3, T, S, S), device=dev)
    qp = torch.as_tensor(synthetic.make_queries(4, Q, T, S, S), device=dev)
    out, fg = tdist.sharded_call(m, video, qp, return_grids=True)
    same = m(tdist.ShapeOnly(video.shape), False, qp, feature_grids=fg)
    bitwise = all(torch.equal(out[k], same[k]) for k in ('tracks', 'occlusion', 'expected_dist'))
    solo = m(video, False, qp)
    d = torch.linalg.norm(out['tracks'] - solo['tracks'], dim=-1)
    shapes_ok = tuple(out['tracks'].shape) == (1, Q, T, 2) and tuple(fg.lowres[0].shape[:2]) == (1, T)
    q.put((rank, backend, bitwise, shapes_ok, float(d.median()), float(d.max()), ''))
    dist.destroy_process_group()
  except Exception as e:   # report instead of hanging the parent
    import traceback
    q.put((rank, '?', False, False, 0.0, 0.0, traceback.format_exc()))


@pytest.mark.parametrize('T,Q', [(6, 10), (7, 5), (1, 3)])   # even, ragged, and an empty frame shard
def test_sharded_call_two_ranks_real_model(T, Q):
  import torch.multiprocessing as mp
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, T, Q, q)) for r in range(2)]
  for p in procs: p.start()
  res = [q.get(timeout=600) for _ in range(2)]
  for p in procs: p.join(60)
  for rank, backend, bitwise, shapes_ok, med, mx, err in res:
    assert not err, err
    assert shapes_ok
    assert bitwise, f'rank {rank} ({backend}): sharded != unsharded on the same grids'
    assert med < 1e-3 and mx < 0.05, (med, mx)
