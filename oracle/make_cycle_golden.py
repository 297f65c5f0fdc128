"""TEST INFRASTRUCTURE ONLY -- pins the cycle-consistency tracker (second half) to the reference's OWN
LINES: tapnet/training/supervised_point_prediction.py:444-531 -- the forward einsum / softmax / heatmaps_to_points,
the vmapped model_utils.interp at the tracked points, the gather of the queries' frames, the backward einsum and its
soft arg max -- are read from the file in /root/reference at run time and EXECUTED over the numpy stand-ins of
oracle/hk_numpy_shim.py (jax / haiku are not installable here; the trainer module itself imports jaxline, optax, tf and
cannot be imported, which is why the lines are lifted out of its method body), together with the reference's
tapnet/utils/model_utils.py and transforms.py imported unmodified.

What is NOT the reference's: line :537, `dist = jnp.square(inverse_tracks - query_points[jnp.newaxis, 2:0:-1])`, slices
the batch axis and cannot broadcast ([b,n,t,2] against [1,k,n,3]) -- it fails for every input under JAX too; the
evident intent, the query's (x, y) = query_points[:, :, None, 2:0:-1], is restated (3 lines below), with the
reference's threshold (48 px) and logits (+10 / -10) of :539-542.

  python oracle/make_cycle_golden.py            # writes tests/golden/cycle_consistency.npz
  python oracle/make_cycle_golden.py --check    # re-runs the reference lines and compares with the committed file
"""
import argparse
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REFERENCE_ROOT = os.environ.get('TAPNET_REFERENCE', '/root/reference')
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden', 'cycle_consistency.npz')
SRC = os.path.join(REFERENCE_ROOT, 'tapnet', 'training', 'supervised_point_prediction.py')
FIRST, LAST = 444, 531     # `feature_grid = output['feature_grid']` ... the closing parenthesis of inverse_tracks

CASES = dict(a=dict(B=1, T=4, hw=(16, 16), Q=9, chunk=4, seed=1),      # ragged query chunks
             b=dict(B=2, T=3, hw=(12, 16), Q=5, chunk=5, seed=2))      # two clips, non-square grid


def l2n(x):
  return (x / np.sqrt(np.maximum(np.sum(np.square(x), -1, keepdims=True), 1e-12))).astype(np.float32)


def reference_function():
  """The reference's lines as a function of (self, output, inputs, input_key) -> (tracks, inverse_tracks)."""
  lines = open(SRC).read().split('\n')[FIRST - 1:LAST]
  assert lines[0].strip() == "feature_grid = output['feature_grid']" and lines[-1].strip() == ')', (lines[0], lines[-1])
  assert 'inverse_tracks = model_utils.heatmaps_to_points(' in '\n'.join(lines[-8:])
  body = textwrap.dedent('\n'.join(lines))
  # the loop body continues in the reference with the distance test; collect what this pin compares instead
  body += '\n  all_tracks.append(tracks)\n  all_inverse.append(inverse_tracks)\n'
  src = ('def cycle(self, output, inputs, input_key):\n  all_inverse = []\n' + textwrap.indent(body, '  ') +
         '\n  return jnp.concatenate(all_tracks, axis=1), jnp.concatenate(all_inverse, axis=1)\n')
  from oracle import hk_numpy_shim as shim
  shim.install()
  if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)
  import jax                                        # the stand-ins
  import jax.numpy as jnp
  from tapnet.utils import model_utils, transforms  # the reference, imported over the stand-ins
  ns = dict(jax=jax, jnp=jnp, np=np, model_utils=model_utils, transforms=transforms)
  exec(compile(src, SRC + f':{FIRST}-{LAST}', 'exec'), ns)
  return ns['cycle']


def run_reference():
  cycle = reference_function()
  out = {}
  for tag, c in CASES.items():
    rng = np.random.default_rng(c['seed'])
    h, w = c['hw']
    H, W = 8 * h, 8 * w
    grid = l2n(rng.standard_normal((c['B'], c['T'], h, w, 256)))
    qp = np.stack([rng.integers(0, c['T'], (c['B'], c['Q'])), rng.uniform(0, H, (c['B'], c['Q'])),
                   rng.uniform(0, W, (c['B'], c['Q']))], -1).astype(np.float32)
    # query features: the grid sampled at the query points (what TAPNet.__call__(get_query_feats=True) returns), so
    # that the forward heat maps have their peak at the query in the query's frame
    from oracle import tapir_oracle as O
    qf = np.stack([O.interp_nearest_3d(grid[b], qp[b] * np.array([1.0, h / H, w / W], np.float32)) for b in range(c['B'])])

    class Self:
      eval_chunk_size = c['chunk']
      softmax_temperature = 10.0
    video = np.zeros((c['B'], c['T'], H, W, 3), np.float32)
    tracks, inverse = cycle(Self(), dict(feature_grid=grid, query_feats=qf), dict(k=dict(query_points=qp, video=video)), 'k')
    # (:537 as intended, :539-542 as written)
    dist = np.sum(np.square(np.asarray(inverse) - qp[:, :, None, 2:0:-1]), axis=-1)
    occlusion = (dist > np.square(48.0)) * 20.0 - 10.0
    out.update({f'{tag}_grid': grid, f'{tag}_query_feats': qf.astype(np.float32), f'{tag}_query_points': qp,
                f'{tag}_tracks': np.asarray(tracks, np.float32), f'{tag}_inverse_tracks': np.asarray(inverse, np.float32),
                f'{tag}_occlusion': occlusion.astype(np.float32), f'{tag}_dist': dist.astype(np.float32),
                f'{tag}_im_hw': np.array([H, W])})
    print(f'[{tag}] tracks {np.asarray(tracks).shape}, occluded {int((occlusion > 0).sum())} of {occlusion.size}')
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--check', action='store_true')
  a = ap.parse_args()
  out = run_reference()
  if a.check:
    old = np.load(OUT)
    worst = max(float(np.abs(old[k].astype(np.float64) - v.astype(np.float64)).max()) for k, v in out.items())
    print(f'committed vs regenerated from the reference lines: max |diff| {worst:.3e}')
    assert set(old.files) == set(out) and worst == 0.0
    return
  np.savez_compressed(OUT, **out)
  print('wrote', OUT, f'{os.path.getsize(OUT) / 1e6:.2f} MB')


if __name__ == '__main__':
  main()
