"""Generates tests/golden/tapvid_metrics.npz by running the REFERENCE's own
compute_tapvid_metrics / sample_queries_* (tapnet/tapvid/evaluation_datasets.py, imported from
/root/reference with its TensorFlow / mediapy imports stubbed: the three functions are pure numpy)
on seeded synthetic tracks.  Run in the build container only:  python oracle/make_tapvid_golden.py"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def import_reference():
  for name in ('tensorflow', 'tensorflow_datasets', 'mediapy', 'absl', 'absl.logging', 'PIL', 'PIL.Image',
               'scipy.io', 'chex'):
    if name not in sys.modules:
      try:
        __import__(name)
      except Exception:
        sys.modules[name] = types.ModuleType(name)
  sys.modules['absl'].logging = sys.modules['absl.logging']
  if not hasattr(sys.modules['chex'], 'Array'):
    sys.modules['chex'].Array = object
    sys.modules['chex'].Shape = object
  tf = sys.modules['tensorflow']
  if not hasattr(tf, 'io'):
    tf.io = types.SimpleNamespace(gfile=types.SimpleNamespace())
  sys.path.insert(0, '/root/reference')
  import importlib.util
  spec = importlib.util.spec_from_file_location('ref_eval', '/root/reference/tapnet/tapvid/evaluation_datasets.py')
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def main():
  ref = import_reference()
  rng = np.random.default_rng(0)
  out = {}
  n, t = 37, 24
  occ = rng.random((n, t)) < 0.3
  occ[5] = True                                  # a track that is never visible
  pts = rng.uniform(0, 256, (n, t, 2))
  frames = rng.uniform(-1, 1, (t, 8, 8, 3)).astype(np.float32)
  for mode, sampler in (('strided', ref.sample_queries_strided), ('first', ref.sample_queries_first)):
    ex = sampler(occ, pts, frames)
    for k in ('query_points', 'target_points', 'occluded'):
      out[f'{mode}_{k}'] = ex[k]
    pred_tracks = ex['target_points'] + rng.normal(0, 3.0, ex['target_points'].shape)
    pred_occ = np.logical_xor(ex['occluded'], rng.random(ex['occluded'].shape) < 0.15)
    out[f'{mode}_pred_tracks'] = pred_tracks
    out[f'{mode}_pred_occ'] = pred_occ
    for trackwise in (False, True):
      m = ref.compute_tapvid_metrics(ex['query_points'], ex['occluded'], ex['target_points'], pred_occ,
                                     pred_tracks, mode, get_trackwise_metrics=trackwise)
      for k, v in m.items():
        out[f'{mode}_{"tw_" if trackwise else ""}{k}'] = np.asarray(v)
  out['occ'] = occ
  out['pts'] = pts
  np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'tapvid_metrics.npz'), **out)
  print('wrote', len(out), 'arrays')


if __name__ == '__main__':
  main()
