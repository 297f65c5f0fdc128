"""TEST INFRASTRUCTURE ONLY -- numpy stand-ins for `jax`, `jax.numpy`, `haiku`, `chex`, `optax` and `absl`,
so that the JAX/Haiku side of the reference can be EXECUTED in the build container (JAX is not installable
here).  Used by ``oracle/make_tapnet_golden.py`` and ``oracle/make_jax_golden.py`` to run the reference's own
``tapnet/models/tapnet_model.py``, ``tapnet/models/tapir_model.py`` (TAPIR / ParameterizedTAPIR),
``tapnet/models/resnet.py`` and ``tapnet/utils/model_utils.py`` line by line; the outputs are committed under
``tests/golden/``.  Nothing under ``tapnet_amd/`` may import this.

What is the reference's and what is this file's:
  * every line of the reference modules runs unmodified: shapes, einsum strings, rearranges, axis choices, the
    order of norms / ReLUs / GELUs, the query-chunk loop with its random permutation, the pyramid, the
    refinement schedule, the causal-context plumbing, WHERE each hk module is constructed and under which
    explicit name;
  * the PRIMITIVES are implemented here from their documented semantics, in float64:
      jnp.*                      -> numpy (same names); arrays carry `.at[idx].set(v)`
      jax.nn.relu/gelu (tanh form, jax's default)/softmax/sigmoid/one_hot, jax.vmap (a Python loop over the
      mapped axes), jax.tree_util.tree_map, jax.lax.stop_gradient, jax.random.permutation (seeded numpy)
      jax.scipy.ndimage.map_coordinates -> scipy.ndimage.map_coordinates (JAX implements scipy's function;
                                            JAX's 'constant' is scipy's 'grid-constant', see install())
      jax.image.resize('bilinear')      -> oracle/jax_resize.py (restatement of jax/_src/image/scale.py:
                                            antialiased when down-sampling)
      hk.Linear                  -> x @ w + b                       (w [in, out])
      hk.Conv2D / hk.Conv3D      -> cross-correlation, channels last, kernel [k.., in, out], padding 'SAME'
                                    with XLA's split of the total padding (low = total // 2): for stride 2 on an
                                    even extent that is (0, 1) -- what the reference's own torch twin states
                                    (tapnet/torch/nets.py:259-263, tapnet/torch/tapir_model.py:747)
      hk.DepthwiseConv1D         -> w [k, 1, C * mult], output channel c * mult + m reads input channel c
                                    (feature_group_count = C), padding 'SAME' or explicit [(lo, hi)]
      hk.LayerNorm / hk.InstanceNorm -> biased variance, eps 1e-5, params `scale` / `offset`
      hk.BatchApply, hk.avg_pool ('VALID', window == stride), hk.remat (identity)
      hk.transform_with_state    -> .init creates the parameters a module asks for, .apply looks them up
  * Haiku's MODULE NAMING is restated too (haiku/_src/module.py unique_and_canonical_name): a module's name
    is `<parent>/<name>` when constructed inside the parent's __call__, `<parent>/~/<name>` inside its
    __init__, `<parent>/~<method>/<name>` inside another method; the default name is the snake-cased class
    name; the second, third ... use of a name in one method call gets `_1`, `_2` ...; explicitly numbered
    names (`block_0`) are kept.  The reference relies on these names itself: its causal state is keyed by
    `hk.experimental.current_name()` and `construct_initial_causal_state` spells the keys out
    (tapir_model.py:1157-1170), so a causal run over this file fails with a KeyError if the rule is wrong.
So the pin is "the reference's code over restated primitives", weaker than a JAX run and stated as such in
DESIGN.md 1.  Initialisers are NOT Haiku's (w_init / b_init are ignored: the ExtraConvs' zero-initialised second
convolution would make the block an identity): every parameter is a seeded normal draw.
"""
import collections
import functools
import re
import sys
import types

import numpy as np

# ------------------------------------------------------------------------------------------ transform state
_T = dict(params=None, init=False, rng=None, stack=[], counters=[])


class Arr(np.ndarray):
  """ndarray with jax's functional update syntax: x.at[idx].set(v)."""

  @property
  def at(self):
    return _At(self)


class _At:
  def __init__(self, a):
    self.a = a

  def __getitem__(self, idx):
    return _AtIdx(self.a, idx)


class _AtIdx:
  def __init__(self, a, idx):
    self.a, self.idx = a, idx

  def set(self, v):
    out = np.array(self.a, copy=True)
    out[self.idx] = v
    return out.view(Arr)


def _arr(x):
  return x.view(Arr) if isinstance(x, np.ndarray) and not isinstance(x, Arr) else x


def _wrap(f):
  @functools.wraps(f)
  def g(*a, **k):
    r = f(*a, **k)
    if isinstance(r, (list, tuple)):
      return type(r)(_arr(v) for v in r)
    return _arr(r)
  return g


# ------------------------------------------------------------------------------------------ module naming
def _snake(name):
  return re.sub(r'((?<=[a-z0-9])[A-Z]|(?!^)[A-Z](?=[a-z]))', r'_\1', name).lower()


def _wrap_method(name, fn):
  @functools.wraps(fn)
  def g(self, *a, **k):
    if not isinstance(self, Module) or '_hk_name' not in self.__dict__:
      return fn(self, *a, **k)      # before Module.__init__ ran (or not a module): nothing to scope
    _T['stack'].append((self, name))
    _T['counters'].append(collections.Counter())
    try:
      return fn(self, *a, **k)
    finally:
      _T['stack'].pop()
      _T['counters'].pop()
  return g


class _ModuleMeta(type):
  def __new__(mcs, cname, bases, ns):
    for k, v in list(ns.items()):
      if isinstance(v, types.FunctionType) and (k == '__call__' or not k.startswith('__')):
        ns[k] = _wrap_method(k, v)
    return super().__new__(mcs, cname, bases, ns)

  def __call__(cls, *a, **k):
    obj = cls.__new__(cls)
    # the module under construction is on the stack with method '__init__' while its constructor runs
    _T['stack'].append((obj, '__init__'))
    _T['counters'].append(collections.Counter())
    try:
      obj.__init__(*a, **k)
    finally:
      _T['stack'].pop()
      _T['counters'].pop()
    return obj


class Module(metaclass=_ModuleMeta):
  def __init__(self, name=None):
    name = name or _snake(type(self).__name__)
    st, ct = _T['stack'], _T['counters']
    assert st and st[-1][0] is self, 'hk.Module.__init__ outside of the constructor protocol'
    if len(st) > 1:
      parent, method = st[-2]
      if method == '__init__':
        name = '~/' + name
      elif method != '__call__':
        name = '~' + method + '/' + name
      name = parent.module_name + '/' + name
    counters = ct[-2] if len(ct) > 1 else _T.setdefault('top', collections.Counter())
    m = re.fullmatch(r'(.*)_(\d+)', name)
    if m:                                   # explicitly numbered: kept, and the counter moves past it
      base, count = m.group(1), int(m.group(2))
      assert counters[base] <= count, f'module name {name} is not unique'
      counters[base] = count + 1
      full = name
    else:
      count = counters[name]
      counters[name] += 1
      full = f'{name}_{count}' if count else name
    self.__dict__['_hk_name'] = full

  @property
  def module_name(self):
    return self.__dict__['_hk_name']

  @property
  def name(self):
    return self.module_name.split('/')[-1]


def get_parameter(name, shape, scale, kind='normal'):
  """Parameter `name` of the innermost module: created (seeded draw, float32) in init mode, looked up in apply."""
  mod = _T['stack'][-1][0].module_name
  store = _T['params']
  if _T['init']:
    if mod not in store or name not in store[mod]:
      r = _T['rng'].standard_normal(shape).astype(np.float32) * np.float32(scale)
      if kind == 'ones':
        r = r + np.float32(1.0)
      store.setdefault(mod, {})[name] = r
  if mod not in store or name not in store[mod]:
    raise KeyError(f'parameter {mod}:{name} not found')
  p = np.asarray(store[mod][name])
  assert tuple(p.shape) == tuple(shape), (mod, name, p.shape, shape)
  return p.astype(np.float64)


def current_name():
  return _T['stack'][-1][0].module_name


class Transformed:
  def __init__(self, f):
    self.f = f

  def _run(self, params, init, rng, a, k):
    saved = dict(_T)
    _T.update(params=params, init=init, rng=np.random.default_rng(rng), stack=[], counters=[],
              top=collections.Counter())
    try:
      return self.f(*a, **k)
    finally:
      _T.clear()
      _T.update(saved)

  def init(self, rng, *a, **k):
    params = {}
    self._run(params, True, int(np.asarray(rng).ravel()[-1]), a, k)
    return params, {}

  def apply(self, params, state, rng, *a, **k):
    out = self._run(params, False, 0 if rng is None else int(np.asarray(rng).ravel()[-1]), a, k)
    return out, (state or {})


# ------------------------------------------------------------------------------------------ layers
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
    x = np.asarray(x, np.float64)
    ci = x.shape[-1]
    w = get_parameter('w', self.kernel_shape + (ci, self.output_channels),
                      1.0 / np.sqrt(ci * np.prod(self.kernel_shape)))
    sp = x.shape[1:-1]
    pads = [_same_pads(n, k, s) for n, k, s in zip(sp, self.kernel_shape, self.stride)]
    xp = np.pad(x, [(0, 0)] + pads + [(0, 0)])
    outs = [-(-n // s) for n, s in zip(sp, self.stride)]
    y = np.zeros((x.shape[0],) + tuple(outs) + (self.output_channels,), np.float64)
    for tap in np.ndindex(*self.kernel_shape):
      sl = tuple(slice(t, t + (o - 1) * s + 1, s) for t, o, s in zip(tap, outs, self.stride))
      y += xp[(slice(None),) + sl] @ w[tap]
    if self.with_bias:
      y = y + get_parameter('b', (self.output_channels,), 0.02)
    return y.view(Arr)


class Conv2D(_ConvND):
  nd = 2


class Conv3D(_ConvND):
  nd = 3


class DepthwiseConv1D(Module):
  def __init__(self, channel_multiplier, kernel_shape, stride=1, rate=1, padding='SAME', with_bias=True,
               w_init=None, b_init=None, data_format='NWC', name=None):
    super().__init__(name)
    assert data_format == 'NWC' and stride == 1 and rate == 1
    self.mult, self.k, self.padding, self.with_bias = channel_multiplier, int(kernel_shape), padding, with_bias

  def __call__(self, x):
    x = np.asarray(x, np.float64)
    c = x.shape[-1]
    w = get_parameter('w', (self.k, 1, c * self.mult), 1.0 / np.sqrt(self.k))
    lo, hi = _same_pads(x.shape[-2], self.k, 1) if self.padding == 'SAME' else tuple(self.padding[0])
    xp = np.pad(x, [(0, 0)] * (x.ndim - 2) + [(lo, hi), (0, 0)])
    n = xp.shape[-2] - self.k + 1
    y = np.zeros(x.shape[:-2] + (n, c * self.mult), np.float64)
    for j in range(self.k):
      y += np.repeat(xp[..., j:j + n, :], self.mult, axis=-1) * w[j, 0]
    if self.with_bias:
      y = y + get_parameter('b', (c * self.mult,), 0.02)
    return y.view(Arr)


class Linear(Module):
  def __init__(self, output_size, with_bias=True, w_init=None, b_init=None, name=None):
    super().__init__(name)
    self.output_size = output_size

  def __call__(self, x):
    x = np.asarray(x, np.float64)
    w = get_parameter('w', (x.shape[-1], self.output_size), 1.0 / np.sqrt(x.shape[-1]))
    return (x @ w + get_parameter('b', (self.output_size,), 0.02)).view(Arr)


class LayerNorm(Module):
  def __init__(self, axis, create_scale, create_offset, eps=1e-5, scale_init=None, offset_init=None,
               use_fast_variance=False, name=None, param_axis=None):
    super().__init__(name)
    assert axis == -1
    self.create_scale, self.create_offset, self.eps = create_scale, create_offset, eps

  def __call__(self, x):
    x = np.asarray(x, np.float64)
    mean = x.mean(-1, keepdims=True)
    var = x.var(-1, keepdims=True)
    y = (x - mean) / np.sqrt(var + self.eps)
    if self.create_scale:
      y = y * get_parameter('scale', (x.shape[-1],), 0.1, 'ones')
    if self.create_offset:
      y = y + get_parameter('offset', (x.shape[-1],), 0.05)
    return y.view(Arr)


class InstanceNorm(Module):
  """hk.InstanceNorm, channels last: statistics over the spatial axes per sample and channel."""

  def __init__(self, create_scale, create_offset, eps=1e-5, scale_init=None, offset_init=None,
               data_format='channels_last', name=None):
    super().__init__(name)
    self.create_scale, self.create_offset, self.eps = create_scale, create_offset, eps

  def __call__(self, x):
    x = np.asarray(x, np.float64)
    ax = tuple(range(1, x.ndim - 1))
    mean = x.mean(ax, keepdims=True)
    var = x.var(ax, keepdims=True)
    y = (x - mean) / np.sqrt(var + self.eps)
    if self.create_scale:
      y = y * get_parameter('scale', (x.shape[-1],), 0.1, 'ones')
    if self.create_offset:
      y = y + get_parameter('offset', (x.shape[-1],), 0.05)
    return y.view(Arr)


class BatchApply(Module):
  def __init__(self, f, num_dims=2, name=None):
    super().__init__(name)
    self.f, self.num_dims = f, num_dims

  def __call__(self, x, *a, **k):
    lead = x.shape[:self.num_dims]
    merge = lambda v: np.reshape(v, (-1,) + v.shape[self.num_dims:])
    split = lambda v: np.reshape(v, lead + v.shape[1:]).view(Arr)
    return tree_map(split, self.f(merge(np.asarray(x)), *a, **k))


def avg_pool(value, window_shape, strides, padding, channel_axis=-1):
  assert padding == 'VALID' and list(window_shape) == list(strides)
  x = np.asarray(value, np.float64)
  crop = tuple(slice(0, (n // w) * w) for n, w in zip(x.shape, window_shape))
  x = x[crop]
  shape, axes = [], []
  for n, w in zip(x.shape, window_shape):
    shape += [n // w, w]
    axes.append(len(shape) - 1)
  return x.reshape(shape).mean(axis=tuple(axes)).view(Arr)


def _unsupported(what):
  class U(Module):
    def __init__(self, *a, name=None, **k):
      super().__init__(name)

    def __call__(self, *a, **k):
      raise NotImplementedError(what + ' is not part of the shim')
  U.__name__ = what
  return U


# ------------------------------------------------------------------------------------------ jax functions
def tree_map(f, tree, *rest):
  if isinstance(tree, dict):
    return {k: tree_map(f, v, *[r[k] for r in rest]) for k, v in tree.items()}
  if isinstance(tree, tuple) and hasattr(tree, '_fields'):
    return type(tree)(*[tree_map(f, v, *[r[i] for r in rest]) for i, v in enumerate(tree)])
  if isinstance(tree, (list, tuple)):
    return type(tree)(tree_map(f, v, *[r[i] for r in rest]) for i, v in enumerate(tree))
  if tree is None:
    return None
  return f(tree, *rest)


def vmap(f, in_axes=0, out_axes=0):
  def g(*args):
    axes = tuple(in_axes) if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
    n = {np.shape(a)[ax] for a, ax in zip(args, axes) if ax is not None}
    assert len(n) == 1
    outs = [f(*[a if ax is None else np.take(a, i, axis=ax) for a, ax in zip(args, axes)])
            for i in range(n.pop())]
    return np.stack(outs, axis=out_axes).view(Arr)
  return g


def _softmax(x, axis=-1):
  z = x - np.max(x, axis=axis, keepdims=True)
  e = np.exp(z)
  return (e / np.sum(e, axis=axis, keepdims=True)).view(Arr)


def _gelu(x, approximate=True):
  assert approximate
  x = np.asarray(x, np.float64)
  return (0.5 * x * (1.0 + np.tanh(np.sqrt(2.0 / np.pi) * (x + 0.044715 * x ** 3)))).view(Arr)


def _one_hot(x, num_classes, axis=-1):
  oh = (np.asarray(x)[..., None] == np.arange(num_classes)).astype(np.float64)
  return np.moveaxis(oh, -1, axis).view(Arr)


def _resize(image, shape, method, antialias=True):
  from oracle import jax_resize
  assert method == 'bilinear' and antialias and len(shape) == 5 and tuple(shape[:2]) == image.shape[:2]
  wy = jax_resize.compute_weight_mat(image.shape[2], int(shape[2]))      # float64 all the way
  wx = jax_resize.compute_weight_mat(image.shape[3], int(shape[3]))
  out = np.einsum('bthwc,hi->btiwc', np.asarray(image, np.float64), wy)
  return np.einsum('btiwc,wj->btijc', out, wx).view(Arr)


def install(params=None):
  """Registers the stand-in modules in sys.modules.  `params`: a Haiku-style dict for direct module use
  outside hk.transform_with_state (apply mode), as oracle/make_tapnet_golden.py does."""
  from scipy import ndimage
  _T.update(params=params, init=False, rng=np.random.default_rng(0), stack=[], counters=[],
            top=collections.Counter())
  jnp = types.ModuleType('jax.numpy')
  for k in dir(np):
    if not k.startswith('_'):
      v = getattr(np, k)
      setattr(jnp, k, _wrap(v) if isinstance(v, (types.FunctionType, types.BuiltinFunctionType, np.ufunc))
              or type(v).__name__ == '_ArrayFunctionDispatcher' else v)
  jnp.array = lambda x, dtype=None: np.array(x, dtype).view(Arr)
  jnp.ndarray = np.ndarray
  nn = types.ModuleType('jax.nn')
  nn.relu = lambda x: np.maximum(x, 0).view(Arr)
  nn.softmax, nn.gelu, nn.one_hot = _softmax, _gelu, _one_hot
  nn.sigmoid = lambda x: (1.0 / (1.0 + np.exp(-np.asarray(x, np.float64)))).view(Arr)
  lax = types.ModuleType('jax.lax')
  lax.stop_gradient = lambda x: x
  jsp = types.ModuleType('jax.scipy')
  jnd = types.ModuleType('jax.scipy.ndimage')
  # JAX's mode='constant' weighs out-of-range TAPS with cval (its docstring: "interpolation near boundaries differs
  # from the scipy function, because we fixed ... scipy/issues/2640"): that is scipy's 'grid-constant', and what the
  # reference's torch twin does with grid_sample(padding_mode='zeros') (tapnet/torch/utils.py:104-118)
  jnd.map_coordinates = lambda x, c, order, mode='constant', cval=0.0: ndimage.map_coordinates(
      np.asarray(x, np.float64), np.asarray(c, np.float64), order=order,
      mode='grid-constant' if mode == 'constant' else mode, cval=cval).view(Arr)
  jsp.ndimage = jnd
  rnd = types.ModuleType('jax.random')
  rnd.PRNGKey = lambda seed: np.array([0, seed], np.uint32)
  rnd.permutation = lambda key, n: np.random.default_rng(int(np.asarray(key).ravel()[-1])).permutation(n).view(Arr)
  tu = types.ModuleType('jax.tree_util')
  tu.tree_map = tree_map
  img = types.ModuleType('jax.image')
  img.resize = _resize
  jax = types.ModuleType('jax')
  jax.jit = lambda f, **k: f                       # tapnet/robotap/tapir_clustering.py:943-952
  jax.custom_vjp = lambda f: f
  jax.Array = np.ndarray
  jax.numpy, jax.nn, jax.lax, jax.scipy, jax.vmap, jax.random, jax.tree_util, jax.image = (
      jnp, nn, lax, jsp, vmap, rnd, tu, img)
  hk = types.ModuleType('haiku')
  hk.Module, hk.Conv2D, hk.Conv3D, hk.Linear, hk.DepthwiseConv1D = Module, Conv2D, Conv3D, Linear, DepthwiseConv1D
  hk.LayerNorm, hk.InstanceNorm, hk.BatchApply, hk.avg_pool = LayerNorm, InstanceNorm, BatchApply, avg_pool
  hk.remat = lambda f: f
  hk.transform_with_state = Transformed
  hk.next_rng_key = lambda: np.array([0, 7], np.uint32)
  hk.Params = hk.State = dict
  hk.data_structures = types.ModuleType('haiku.data_structures')
  hk.data_structures.tree_size = lambda t: sum(int(np.size(v)) for m in t.values() for v in m.values())
  hk.data_structures.tree_bytes = lambda t: sum(int(np.asarray(v).nbytes) for m in t.values() for v in m.values())
  hk.experimental = types.ModuleType('haiku.experimental')
  hk.experimental.current_name = current_name
  for n in ('BatchNorm', 'MaxPool'):
    setattr(hk, n, _unsupported(n))
  hk.max_pool = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError('max_pool'))
  chex = types.ModuleType('chex')
  chex.Array, chex.Shape, chex.Numeric, chex.PRNGKey = np.ndarray, tuple, float, np.ndarray
  absl = types.ModuleType('absl')
  absl.logging = types.ModuleType('absl.logging')
  absl.logging.info = absl.logging.warning = lambda *a, **k: None
  optax = types.ModuleType('optax')
  optax.OptState = object
  for name, mod in (('jax', jax), ('jax.numpy', jnp), ('jax.nn', nn), ('jax.lax', lax), ('jax.scipy', jsp),
                    ('jax.scipy.ndimage', jnd), ('jax.random', rnd), ('jax.tree_util', tu), ('jax.image', img),
                    ('haiku', hk), ('haiku.experimental', hk.experimental), ('chex', chex), ('absl', absl),
                    ('absl.logging', absl.logging), ('optax', optax)):
    sys.modules[name] = mod
