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


def test_bench_line_carries_per_class_rooflines_and_the_rank_projection():
  """One GPU: the line of `bench.py --emulate-rank 2` has `roofline_all` (one roofline entry per kernel class, backbone
  classes from the eager profiling pass) and `emulated_ranks` (one rank's measured share + the priced exchange,
  labelled a projection), next to the contract's fields."""
  line = _run(['--steps', '3', '--warmup', '1', '--no-accuracy', '--no-cpu-baseline', '--emulate-rank', '2'])
  assert line['n_gpus'] == 1 and line['dtype'] == 'bf16' and line['value'] > 0 and line['vs_baseline'] is None
  ra = line['roofline_all']
  assert {'mixer_fused', 'cv_rows', 'patch_corr', 'conv3x3_c64', 'conv3x3_c128', 'conv3x3_c256', 'conv_other', 'stem',
          'l2norm'} <= set(ra)
  for k, v in ra.items():
    assert v['bound'] in ('mfma', 'hbm') and 0 < v['frac'] < 1.2 and v['launches'] > 0, (k, v)
  assert ra['conv3x3_c256']['launches'] == 7 and ra['conv_other']['launches'] == 2      # dual launches: 19 per frame group
  er = line['emulated_ranks']
  assert 'PROJECTION' in er['what'] and er['exchange']['wire'].startswith('bf16') and len(er['ranks']) == 1
  r = er['ranks'][0]
  assert r['world'] == 2 and r['frames_per_rank'] == 24 and r['queries_per_rank'] == 128
  assert abs(r['per_rank_ms'] - (r['backbone_ms'] + r['hot_path_ms'] + r['exchange_ms_priced'])) < 2e-3
  assert 'roofline' in line and line['roofline']['frac'] > 0.2 and 'box' in line
