"""TEST INFRASTRUCTURE ONLY -- numpy stand-ins for `jax`, `jax.numpy`, `haiku`, `chex`, `optax` and `absl`,
so that the JAX/Haiku side of the reference that has NO torch twin can be EXECUTED in the build container
(JAX is not installable here).  Used by ``oracle/make_tapnet_golden.py`` to run the reference's own
``tapnet/models/tapnet_model.py`` (TAPNet.__call__ / tracks_from_cost_volume) and the reference's own
``tapnet/utils/model_utils.py`` (heatmaps_to_points, soft_argmax_heatmap, interp) line by line; the outputs
are committed as ``tests/golden/tapnet_head.npz``.  Nothing under ``tapnet_amd/`` may import this.

What is the reference's and what is this file's:
  * every line of tapnet_model.py / model_utils.py / transforms.py runs unmodified (shapes, einsum strings,
    rearranges, axis choices, the order of ReLUs, the chunk loop, the query-point override);
  * the PRIMITIVES they call are implemented here from their documented semantics:
      jnp.*                      -> numpy (same names; float64 where numpy promotes)
      jax.nn.relu/softmax/sigmoid, jax.vmap (a Python loop over the mapped axes), jax.lax.stop_gradient
      jax.scipy.ndimage.map_coordinates -> scipy.ndimage.map_coordinates (JAX implements scipy's function)
      hk.Linear                  -> x @ w + b                       (w [in, out])
      hk.Conv3D / hk.Conv2D      -> cross-correlation, channels last, kernel [k.., in, out], padding 'SAME'
                                    with XLA's split of the total padding (low = total // 2): for stride 2 on an
                                    even extent that is (0, 1) -- what the reference's own torch twin states
                                    for the same layer of TAPIR (tapnet/torch/nets.py:259-263,
                                    tapnet/torch/tapir_model.py:747)
    Parameters come from a Haiku-style dict {'<scope>/<module name>': {'w': ..., 'b': ...}} handed to
    `install(params)`; modules look themselves up by name.
So the pin is "the reference's code over restated primitives", weaker than a JAX run and stated as such in
DESIGN.md 1.
"""
import sys
import types

import numpy as np

_PARAMS = {}
_SCOPE = ['']


def _lookup(name):
  hits = [k for k in _PARAMS if k == name or k.endswith('/' + name)]
  if len(hits) != 1:
    raise KeyError(f'{name}: {len(hits)} parameter entries')
  return _PARAMS[hits[0]]


class Module:
  def __init__(self, name=None):
    self.name = name or type(self).__name__


def _same_pads(n, k, s):
  out = -(-n // s)
  total = max((out - 1) * s + k - n, 0)
  return total // 2, total - total // 2


class _ConvND(Module):
  nd = 2

  def __init__(self, output_channels, kernel_shape, stride=1, rate=1, padding='SAME', with_bias=True,
               w_init=None, b_init=None, data_format=None, mask=None, feature_group_count=1, name=None):
    super().__init__(name)
    assert padding == 'SAME' and rate == 1 and feature_group_count == 1 and mask is None
    seq = lambda v: tuple(v) if isinstance(v, (list, tuple)) else (v,) * self.nd
    self.output_channels, self.kernel_shape, self.stride = output_channels, seq(kernel_shape), seq(stride)
    self.with_bias = with_bias

  def __call__(self, x):
    p = _lookup(self.name)
    w = np.asarray(p['w'], np.float64)
    assert w.shape == self.kernel_shape + (x.shape[-1], self.output_channels), (w.shape, x.shape)
    x = np.asarray(x, np.float64)
    sp = x.shape[1:-1]
    pads = [_same_pads(n, k, s) for n, k, s in zip(sp, self.kernel_shape, self.stride)]
    xp = np.pad(x, [(0, 0)] + pads + [(0, 0)])
    outs = [-(-n // s) for n, s in zip(sp, self.stride)]
    y = np.zeros((x.shape[0],) + tuple(outs) + (self.output_channels,), np.float64)
    for tap in np.ndindex(*self.kernel_shape):
      sl = tuple(slice(t, t + (o - 1) * s + 1, s) for t, o, s in zip(tap, outs, self.stride))
      y += xp[(slice(None),) + sl] @ w[tap]
    if self.with_bias:
      y = y + np.asarray(p['b'], np.float64)
    return y


class Conv2D(_ConvND):
  nd = 2


class Conv3D(_ConvND):
  nd = 3


class Linear(Module):
  def __init__(self, output_size, with_bias=True, w_init=None, b_init=None, name=None):
    super().__init__(name)
    self.output_size = output_size

  def __call__(self, x):
    p = _lookup(self.name)
    w = np.asarray(p['w'], np.float64)
    assert w.shape == (x.shape[-1], self.output_size)
    return np.asarray(x, np.float64) @ w + np.asarray(p['b'], np.float64)


def _unsupported(what):
  class U(Module):
    def __init__(self, *a, name=None, **k):
      super().__init__(name)

    def __call__(self, *a, **k):
      raise NotImplementedError(what + ' is not part of the shim (the TSM-ResNet backbone is out of scope)')
  U.__name__ = what
  return U


def vmap(f, in_axes=0, out_axes=0):
  def g(*args):
    axes = tuple(in_axes) if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
    n = {np.shape(a)[ax] for a, ax in zip(args, axes) if ax is not None}
    assert len(n) == 1
    outs = [f(*[a if ax is None else np.take(a, i, axis=ax) for a, ax in zip(args, axes)])
            for i in range(n.pop())]
    return np.stack(outs, axis=out_axes)
  return g


def _softmax(x, axis=-1):
  z = x - np.max(x, axis=axis, keepdims=True)
  e = np.exp(z)
  return e / np.sum(e, axis=axis, keepdims=True)


def install(params):
  """Registers the stand-in modules in sys.modules and the Haiku-style parameter dict."""
  from scipy import ndimage
  _PARAMS.clear()
  _PARAMS.update(params)
  jnp = types.ModuleType('jax.numpy')
  for k in dir(np):
    if not k.startswith('_'):
      setattr(jnp, k, getattr(np, k))
  jnp.array = lambda x, dtype=None: np.asarray(x, dtype)
  nn = types.ModuleType('jax.nn')
  nn.relu = lambda x: np.maximum(x, 0)
  nn.softmax = _softmax
  nn.sigmoid = lambda x: 1.0 / (1.0 + np.exp(-x))
  lax = types.ModuleType('jax.lax')
  lax.stop_gradient = lambda x: x
  jsp = types.ModuleType('jax.scipy')
  jnd = types.ModuleType('jax.scipy.ndimage')
  jnd.map_coordinates = lambda x, c, order, mode='constant', cval=0.0: ndimage.map_coordinates(
      np.asarray(x, np.float64), np.asarray(c, np.float64), order=order, mode=mode, cval=cval)
  jsp.ndimage = jnd
  jax = types.ModuleType('jax')
  jax.numpy, jax.nn, jax.lax, jax.scipy, jax.vmap = jnp, nn, lax, jsp, vmap
  hk = types.ModuleType('haiku')
  hk.Module, hk.Conv2D, hk.Conv3D, hk.Linear = Module, Conv2D, Conv3D, Linear
  for n in ('BatchNorm', 'MaxPool', 'LayerNorm', 'InstanceNorm'):
    setattr(hk, n, _unsupported(n))
  chex = types.ModuleType('chex')
  chex.Array, chex.Shape, chex.Numeric, chex.PRNGKey = np.ndarray, tuple, float, np.ndarray
  absl = types.ModuleType('absl')
  absl.logging = types.ModuleType('absl.logging')
  absl.logging.info = absl.logging.warning = lambda *a, **k: None
  optax = types.ModuleType('optax')
  for name, mod in (('jax', jax), ('jax.numpy', jnp), ('jax.nn', nn), ('jax.lax', lax), ('jax.scipy', jsp),
                    ('jax.scipy.ndimage', jnd), ('haiku', hk), ('chex', chex), ('absl', absl),
                    ('absl.logging', absl.logging), ('optax', optax)):
    sys.modules[name] = mod
