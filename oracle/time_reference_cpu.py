"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- times the REFERENCE's own PyTorch TAPIR on this
host's CPU cores at the headline workload (BASELINE.json configs[1]: 256x256x48 clip, 256
queries), SURVEY.md 8d "CPU baseline procedure": fp32, all cores, 1 warm-up + 3 timed runs,
median.  The JAX path cannot run offline (no jax wheel); the reference's torch twin
(tapnet/torch/tapir_model.py) is the reference's CPU path that exists here.

    python -m oracle.time_reference_cpu [--model tapir|bootstapir] [--runs 3]

Needs /root/reference, so it runs in the build container only; the result is committed under
profiles/ and bench.py quotes it as ``cpu_baseline.reference_torch`` (with the host it was
measured on), next to the port it times live on the GPU box.
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_import import import_reference  # noqa: E402
from tapnet_amd import synthetic  # noqa: E402

MODELS = {
    'tapir': dict(pyramid_level=0, extra_convs=False, softmax_temperature=20.0),
    'bootstapir': dict(pyramid_level=1, extra_convs=True, softmax_temperature=10.0),
}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--model', default='tapir', choices=list(MODELS))
  ap.add_argument('--frames', type=int, default=48)
  ap.add_argument('--queries', type=int, default=256)
  ap.add_argument('--size', type=int, default=256)
  ap.add_argument('--runs', type=int, default=3)
  ap.add_argument('--out', default=None)
  args = ap.parse_args()
  cores = os.cpu_count() or 1
  torch.set_num_threads(cores)
  tm, _, _ = import_reference()
  kw = MODELS[args.model]
  weights = synthetic.make_weights(0, kw['pyramid_level'], kw['extra_convs'])
  model = tm.TAPIR(pyramid_level=kw['pyramid_level'], extra_convs=kw['extra_convs'],
                   softmax_temperature=kw['softmax_temperature'],
                   initial_resolution=(args.size, args.size)).eval()
  model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()}, strict=True)
  video = torch.from_numpy(synthetic.make_video(1, args.frames, args.size, args.size))
  qpts = torch.from_numpy(synthetic.make_queries(101, args.queries, args.frames, args.size, args.size))
  times = []
  with torch.no_grad():
    for i in range(args.runs + 1):
      t0 = time.perf_counter()
      out = model(video, qpts)   # torch twin defaults: query_chunk_size=64
      dt = time.perf_counter() - t0
      assert torch.isfinite(out['tracks']).all()
      if i:
        times.append(dt)
      print(f'run {i}: {dt:.2f} s', flush=True)
  med = float(np.median(times))
  cpu = ''
  try:
    for line in open('/proc/cpuinfo'):
      if line.startswith('model name'):
        cpu = line.split(':', 1)[1].strip()
        break
  except OSError:
    pass
  res = dict(what='reference tapnet/torch/tapir_model.py TAPIR.forward on CPU (torch %s)' % torch.__version__,
             model=args.model, workload=f'{args.size}x{args.size}x{args.frames} clip, Q={args.queries}',
             cores=cores, cpu=cpu or platform.processor(), runs=times, median_s=round(med, 3),
             points_per_s=round(args.queries / med, 3), host='build container (not the GPU box)')
  print(json.dumps(res))
  if args.out:
    with open(args.out, 'w') as f:
      json.dump(res, f, indent=1)


if __name__ == '__main__':
  main()
