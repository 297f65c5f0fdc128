"""Seeded random-shape parity sweep (round 3's tools/fuzz_parity.py runs, now a test): the f32 engine, video -> tracks,
against the oracle (plain-PyTorch backbone restatement + numpy hot path) on random configurations -- frame counts,
query counts, NON-SQUARE initial resolutions, frames at 1x / 1.5x / 2x of it (2 or 3 feature levels), TAPIR / BootsTAPIR
kwargs, ragged query chunks.  Sweep A: small shapes (separate-launch kernels); sweep B: up to 50 frames x 96 queries
(the track-resident mixer kernels engage).

What is asserted, per sweep:
  * the margin mask is REPORTED and small: a query is set aside only when one of its frames has a relative gap below
    1e-4 between the two largest soft-max cells of the oracle's heat map (a near-tie arg max may legitimately flip
    between two f32 evaluation orders, and the temporal convolutions spread the flip over the query's frames);
    at least 90 % of the queries must remain;
  * on the remaining queries every point, every frame, PER CASE: tracks within 1e-3 x (video / initial resolution) px
    in VIDEO pixels -- i.e. within 1e-3 px of the initial-resolution estimate -- and logits within 1e-3.  The engine's
    tracks are in video pixels (tapir_model.py:906-912 rescales by video / initial resolution), so on a frame at twice
    the initial resolution a deviation of 5.7e-4 px of the 256-coordinate estimate reads 1.14e-3 px (round 3's sweep,
    profiles/r03_fuzz_parity.txt); north_star states its tolerance for video = initial resolution, and the re-scaling
    is in the ASSERTION (`tracks_video_px_max <= 1e-3 * scale` below), not only here.  Both numbers are recorded
    (gpurun_out/fuzz_parity.json).  40 + 8 cases (round 5; 10 + 2 before).
Reference arithmetic: tapnet/models/tapir_model.py:626-729, 858-1154; tapnet/utils/model_utils.py:209-314."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import backbone_torch, tapir_oracle as O   # checker only
from tapnet_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEPS = {'A_small': dict(seed=1, cases=40, tmax=20, qmax=24), 'B_fused_mixer': dict(seed=7, cases=8, tmax=50, qmax=96)}


def _case(rng, tmax, qmax):
  pyr, extra = int(rng.integers(0, 2)), bool(rng.integers(0, 2))
  T, Q = int(rng.integers(1, tmax + 1)), int(rng.integers(1, qmax + 1))
  ih, iw = 8 * int(rng.integers(6, 13)), 8 * int(rng.integers(6, 13))          # initial_resolution 48..96
  scale = float(rng.choice([1.0, 1.0, 1.5, 2.0]))
  H, W = 8 * int(round(ih * scale / 8)), 8 * int(round(iw * scale / 8))
  chunk = int(rng.choice([Q, max(1, Q // 2), 7]))
  seed = int(rng.integers(0, 1 << 30))
  return dict(pyr=pyr, extra=extra, T=T, Q=Q, init=(ih, iw), video=(H, W), chunk=chunk, seed=seed)


def _oracle_case(c):
  """The CPU side of one case (plain-PyTorch backbone restatement + numpy hot path), from the case's seeds alone: runs
  in a worker process next to the GPU calls of the other cases (the oracle is 90 % of the sweep's time)."""
  torch.set_num_threads(2)
  (ih, iw), (H, W), T, Q = c['init'], c['video'], c['T'], c['Q']
  w = synthetic.make_weights(c['seed'] % 1000, c['pyr'], c['extra'])
  video = synthetic.make_video(c['seed'], T, H, W).astype(np.float32)
  qp = synthetic.make_queries(c['seed'] + 1, Q, T, H, W).astype(np.float32)
  res = [(ih, iw)] + [tuple(r) for r in O.generate_default_resolutions((H, W), (ih, iw))]
  bb = backbone_torch.TorchBackbone(w, c['extra'])
  lows, his, cur, lo, hi = [], [], None, None, None
  for r in res:
    if r != cur:
      v = torch.as_tensor(video)
      if r != (H, W):   # the torch twin's resize (no antialias): what the engine does for torch-named weights
        v = torch.nn.functional.interpolate(v[0].permute(0, 3, 1, 2), size=r, mode='bilinear', align_corners=False
                                            ).permute(0, 2, 3, 1)[None]
      l, h = bb.features(v.reshape(-1, r[0], r[1], 3))
      lo, hi, cur = l.numpy()[None], h.numpy()[None], r
    lows.append(lo); his.append(hi)
  ref = O.tapir_from_grids(w, video.shape, lows, his, res, qp, pyramid_level=c['pyr'], softmax_temperature=20.0,
                           initial_resolution=(ih, iw))
  ql, _ = O.get_query_features(lows, his, res, qp, video.shape)
  _, _, _, st = O.tracks_from_cost_volume(w, ql[0], lows[0], None, (ih, iw), 20.0, return_stages=True)
  return dict(tracks=ref['tracks'], occlusion=ref['occlusion'], expected_dist=ref['expected_dist'], levels=len(res),
              top2_rel_gap=st['top2_rel_gap'])


def _oracle_worker_main(cases_path, out_dir, start, step):
  """Child interpreter: the oracle of cases start, start + step, ... -> out_dir/ref_<i>.npz (written under a temporary
  name first: a file that exists is complete)."""
  cases = json.load(open(cases_path))
  for i in range(start, len(cases), step):
    c = cases[i]
    c['init'], c['video'] = tuple(c['init']), tuple(c['video'])
    r = _oracle_case(c)
    tmp = os.path.join(out_dir, f'tmp_{i}.npz')
    np.savez(tmp, **r)
    os.replace(tmp, os.path.join(out_dir, f'ref_{i}.npz'))


@pytest.mark.parametrize('sweep', list(SWEEPS))
def test_random_shapes_against_the_oracle(sweep):
  import subprocess
  import sys
  import tempfile
  from tapnet_amd import tapir_model
  cfg = SWEEPS[sweep]
  rng = np.random.default_rng(cfg['seed'])
  cases = [_case(rng, cfg['tmax'], cfg['qmax']) for _ in range(cfg['cases'])]
  rows, n_q, n_clear = [], 0, 0
  # The oracle is 90 % of this test's time and plain single-threaded numpy for the most part: its cases run in four fresh
  # child interpreters (no GPU; 4 BLAS / torch threads each: the GPU boxes give 16 CPUs of quota) next to the engine's
  # calls here.
  nproc = 4
  with tempfile.TemporaryDirectory() as td:
    cpath = os.path.join(td, 'cases.json')
    json.dump(cases, open(cpath, 'w'))
    env = dict(os.environ, OMP_NUM_THREADS='4', MKL_NUM_THREADS='4', OPENBLAS_NUM_THREADS='4', HIP_VISIBLE_DEVICES='')
    code = ('import sys; sys.path.insert(0, sys.argv[1]); from tests.test_gpu_fuzz_parity import _oracle_worker_main; '
            '_oracle_worker_main(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))')
    kids = [subprocess.Popen([sys.executable, '-c', code, ROOT, cpath, td, str(k), str(nproc)], env=env,
                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for k in range(nproc)]
    try:
      outs = []
      for c in cases:
        (ih, iw), (H, W), T, Q = c['init'], c['video'], c['T'], c['Q']
        w = synthetic.make_weights(c['seed'] % 1000, c['pyr'], c['extra'])
        video = synthetic.make_video(c['seed'], T, H, W).astype(np.float32)
        qp = synthetic.make_queries(c['seed'] + 1, Q, T, H, W).astype(np.float32)
        m = tapir_model.TAPIR(pyramid_level=c['pyr'], extra_convs=c['extra'], initial_resolution=(ih, iw), weights=w,
                              device='cuda:0')
        out = m(video, False, qp, query_chunk_size=c['chunk'])
        outs.append({k: np.asarray(out[k]) for k in ('tracks', 'occlusion', 'expected_dist')})
        del m
      for k, p in enumerate(kids):
        _, err = p.communicate(timeout=1500)
        assert p.returncode == 0, f'oracle worker {k} failed: ' + err.decode(errors='replace')[-2000:]
    finally:
      for p in kids:
        if p.poll() is None:
          p.kill()
    refs = [dict(np.load(os.path.join(td, f'ref_{i}.npz'))) for i in range(len(cases))]
  for i, (c, out, ref) in enumerate(zip(cases, outs, refs)):
    (ih, iw), (H, W), T, Q = c['init'], c['video'], c['T'], c['Q']
    clear = (ref['top2_rel_gap'] > 1e-4).all(axis=-1)[0]          # [Q]: no near-tie arg max in any frame of the query
    d = np.linalg.norm(out['tracks'] - ref['tracks'], axis=-1)[0]                    # [Q, T] video px
    dl = np.maximum(np.abs(out['occlusion'] - ref['occlusion']),
                    np.abs(out['expected_dist'] - ref['expected_dist']))[0]
    scale = max(H / ih, W / iw)
    row = dict(case=i, **{k: c[k] for k in ('pyr', 'extra', 'T', 'Q', 'init', 'video', 'chunk')}, levels=int(ref['levels']),
               queries_masked=int((~clear).sum()),
               tracks_video_px_max=float(d[clear].max()) if clear.any() else 0.0,
               tracks_initial_px_max=float(d[clear].max() / scale) if clear.any() else 0.0,
               logits_max=float(dl[clear].max()) if clear.any() else 0.0,
               masked_tracks_video_px_max=float(d[~clear].max()) if (~clear).any() else 0.0,
               min_top2_rel_gap=float(ref['top2_rel_gap'].min()))
    rows.append(row)
    # the bound in video pixels, per case: 1e-3 px of the initial-resolution estimate, re-scaled as the model re-scales
    assert row['tracks_video_px_max'] <= 1e-3 * scale and row['logits_max'] < 1e-3, row
    n_q += Q; n_clear += int(clear.sum())
  frac = n_clear / n_q
  summary = dict(sweep=sweep, **cfg, queries=n_q, fraction_compared=round(frac, 4),
                 tracks_initial_px_max=max(r['tracks_initial_px_max'] for r in rows),
                 tracks_video_px_max=max(r['tracks_video_px_max'] for r in rows),
                 logits_max=max(r['logits_max'] for r in rows), case_rows=rows)
  os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
  path = os.path.join(ROOT, 'gpurun_out', 'fuzz_parity.json')
  allr = json.load(open(path)) if os.path.exists(path) else {}
  allr[sweep] = summary
  json.dump(allr, open(path, 'w'), indent=1)
  print(json.dumps({k: v for k, v in summary.items() if k != 'case_rows'}))
  assert frac >= 0.9, f'margin mask hides {100 * (1 - frac):.1f} % of the queries'
  assert summary['tracks_initial_px_max'] < 1e-3 and summary['logits_max'] < 1e-3, summary
