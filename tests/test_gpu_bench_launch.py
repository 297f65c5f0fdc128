"""`-m gpu`: `python bench.py --gpus 2` starts its own ranks (the driver's N > 1 contract can also be met
without an external launcher).  On a box with >= 2 devices the ranks take one GPU each over RCCL; on the
one-GPU test box both ranks share the device and the collectives fall back to gloo (RCCL refuses two
ranks on one device) -- correctness of the launch / reduce / sharded path only, no performance claim."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
  env = dict(os.environ)
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
    env.pop(k, None)
  p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + extra, cwd=ROOT, env=env,
                     capture_output=True, text=True, timeout=900)
  assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
  lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, p.stdout[-2000:]
  return json.loads(lines[0])


def test_bench_gpus_2_self_launch():
  line = _run(['--gpus', '2', '--steps', '2', '--warmup', '1'])
  assert line['n_gpus'] == 2 and line['steps'] == 2 and line['scaling'] == 'weak'
  assert line['config']['clips'] == 2 and line['value'] > 0
  assert abs(line['value'] - 2 * 256 / (line['ms_per_step'] * 1e-3)) / line['value'] < 1e-3
  sh = line['one_clip_sharded']
  assert sh['scaling'] == 'strong' and sh['value'] > 0 and 'bf16 on the wire' in sh['exchange']
  if torch.cuda.device_count() >= 2:
    assert line['config']['backend'] == 'nccl' and line['rccl_ranks'] == 2
  else:
    assert line['config']['backend'] == 'gloo' and line['rccl_ranks'] == 0
