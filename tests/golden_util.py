"""Helpers shared by the CPU (oracle) and GPU (HIP) parity tests."""
import os

import numpy as np

from tapnet_amd import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# must match oracle/make_golden.py:CASES
CASES = {
    'tapir': dict(pyramid_level=0, extra_convs=False, softmax_temperature=20.0,
                  causal=False, res=64, video=64, T=5, Q=10, wseed=11),
    'bootstapir': dict(pyramid_level=1, extra_convs=True, softmax_temperature=10.0,
                       causal=False, res=64, video=64, T=5, Q=10, wseed=12),
    'causal': dict(pyramid_level=1, extra_convs=False, softmax_temperature=20.0,
                   causal=True, res=64, video=64, T=4, Q=6, wseed=13),
    'multires': dict(pyramid_level=1, extra_convs=False, softmax_temperature=20.0,
                     causal=False, res=64, video=128, T=3, Q=6, wseed=14),
    'causal_update': dict(pyramid_level=1, extra_convs=False, softmax_temperature=20.0,
                          causal=True, res=64, video=128, T=5, Q=6, wseed=15,
                          update_frame=2, update_idx=(1, 4)),
}


def load_case(name):
  cfg = CASES[name]
  g = dict(np.load(os.path.join(GOLDEN_DIR, name + '.npz')))
  src = g['level_src']
  g['lowres'] = [g[f'lowres_{s}'] for s in src]
  g['hires'] = [g[f'hires_{s}'] for s in src]
  g['qlowres'] = [g[f'qlowres_{s}'] for s in src]
  g['qhires'] = [g[f'qhires_{s}'] for s in src]
  g['res_list'] = [tuple(int(v) for v in r) for r in g['resolutions']]
  weights = synthetic.make_weights(cfg['wseed'], cfg['pyramid_level'], cfg['extra_convs'])
  return cfg, g, weights


def oracle_kwargs(cfg):
  return dict(num_pips_iter=4, pyramid_level=cfg['pyramid_level'],
              softmax_temperature=cfg['softmax_temperature'],
              initial_resolution=(cfg['res'], cfg['res']),
              use_causal_conv=cfg['causal'])


# must match oracle/make_golden.py:HEADLINE -- the benchmarked shape (BASELINE.json configs[1]), the reference's
# torch twin end to end; the fixtures hold OUTPUTS only, inputs and weights are regenerated from the seeds
HEADLINE = {
    'headline_tapir': dict(pyramid_level=0, extra_convs=False, softmax_temperature=20.0, causal=False,
                           res=256, video=256, T=48, Q=256, wseed=31),
    'headline_bootstapir': dict(pyramid_level=1, extra_convs=True, softmax_temperature=10.0, causal=False,
                                res=256, video=256, T=48, Q=256, wseed=32),
}


def load_headline(name):
  cfg = HEADLINE[name]
  g = dict(np.load(os.path.join(GOLDEN_DIR, name + '.npz')))
  weights = synthetic.make_weights(cfg['wseed'], cfg['pyramid_level'], cfg['extra_convs'])
  video = synthetic.make_video(cfg['wseed'] + 100, cfg['T'], cfg['video'], cfg['video'])
  qpts = synthetic.make_queries(cfg['wseed'] + 200, cfg['Q'], cfg['T'], cfg['video'], cfg['video'])
  return cfg, g, weights, video, qpts
