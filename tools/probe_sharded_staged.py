"""Two ranks on one GPU (gloo): data-level checks of the staged gather (GPU probe for tests/test_gpu_distributed.py)."""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def worker(rank, world, port, q):
  try:
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    from tapnet_amd import distributed as tdist, synthetic, tapir_model
    dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    S, T, Q = 64, 9, 10
    w = synthetic.make_weights(17, pyramid_level=1, extra_convs=False)
    m = tapir_model.TAPIR(pyramid_level=1, weights=w, device=dev, initial_resolution=(S, S), dtype='bfloat16')
    video = torch.as_tensor(synthetic.make_video(3, T, S, S), device=dev)
    qp = torch.as_tensor(synthetic.make_queries(4, Q, T, S, S), device=dev)
    t0, t1 = tdist.shard_range(T, world, rank)
    fg = tdist.gather_feature_grids(m, video[:, t0:t1], T, None, torch.bfloat16)
    msgs = [f'rank {rank}: type {type(fg).__name__}, staged {len(getattr(fg, "staged", []) or [])}']
    (lo32, lo16, lot), (hi32, hi16, _) = fg.staged
    msgs.append(f'f32 == bf16 copy: low {bool(torch.equal(lo32, lo16.float()))} hi {bool(torch.equal(hi32, hi16.float()))}; '
                f'fg.lowres[0] is staged key: {fg.lowres[0].data_ptr() == lo32.data_ptr()}')
    tl = lot.reshape(T, -1, 32, 16, 8)
    rm = lo16.reshape(T, -1, 16, 32, 8).permute(0, 1, 3, 2, 4).contiguous()
    bad = [int(t) for t in range(T) if not torch.equal(tl[t], rm[t])]
    msgs.append(f'tile-order copy == re-tiled row-major: frames that differ {bad}; shapes {tuple(lot.shape)} {tuple(lo16.shape)}')
    # against the whole clip computed locally
    m._staged = []
    full = m.get_feature_grids(video, _borrow=True)
    st, m._staged = list(m._staged), []
    msgs.append(f'gathered low16 == local whole-clip low16: {bool(torch.equal(lo16.reshape(st[0][1].shape), st[0][1]))}; '
                f'tiled: {bool(torch.equal(lot.reshape(st[0][2].shape), st[0][2]))}; hi16: {bool(torch.equal(hi16.reshape(st[1][1].shape), st[1][1]))}')
    keys = ('tracks', 'occlusion', 'expected_dist')
    sh = tdist.ShapeOnly(video.shape)
    q0, q1 = tdist.shard_range(Q, world, rank)
    plain = tapir_model.FeatureGrids(fg.lowres, fg.hires, fg.resolutions)
    a = [m(sh, False, qp[:, q0:q1], feature_grids=fg) for _ in range(2)]
    b = [m(sh, False, qp[:, q0:q1], feature_grids=plain) for _ in range(2)]
    dm = lambda x, y: max(float((x[k] - y[k]).abs().max()) for k in keys)
    msgs.append(f'staged twice {dm(a[0], a[1])}, plain twice {dm(b[0], b[1])}, staged vs plain {dm(a[0], b[0])}')
    q.put('\n'.join(msgs))
    dist.destroy_process_group()
  except Exception:
    import traceback
    q.put(f'rank {rank}: ' + traceback.format_exc())


if __name__ == '__main__':
  import torch.multiprocessing as mp
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
  for p in ps: p.start()
  for _ in ps: print(q.get(timeout=600))
  for p in ps: p.join(60)
