"""TEST INFRASTRUCTURE ONLY -- import shim for the *reference* PyTorch TAPIR.

Only usable in the build container, where /root/reference exists.  It is used
by ``oracle/make_golden.py`` to generate the committed fixtures under
``tests/golden/`` and by a few CPU tests (skipped when the reference tree is
absent, e.g. on the GPU box).  Nothing in ``tapnet_amd/`` may import this.

The reference torch model (tapnet/torch/tapir_model.py) needs two un-installed
third-party modules:
  * ``einshape``  (tapnet/torch/utils.py:19-20)  -> stubbed, and
    ``tapnet.torch.utils.einshape`` is replaced by an einops adapter
  * ``tree`` (dm-tree) (tapnet/torch/tapir_model.py:27) -> ``map_structure`` stub
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("TAPNET_REFERENCE", "/root/reference")
# the GPU box has no /root/reference: oracle/stage_ref.py (run by __graft_entry__.build() in the build container)
# stages the reference's torch twin into the git-ignored oracle/_ref/, which travels with the push
STAGED_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")
if not os.path.isdir(os.path.join(REFERENCE_ROOT, "tapnet", "torch")) and \
    os.path.isfile(os.path.join(STAGED_ROOT, "tapnet", "torch", "tapir_model.py")):
  REFERENCE_ROOT = STAGED_ROOT


def reference_available() -> bool:
  return os.path.isdir(os.path.join(REFERENCE_ROOT, "tapnet", "torch"))


def reference_is_staged_copy() -> bool:
  return REFERENCE_ROOT == STAGED_ROOT


def _einshape_adapter():
  import einops

  def einshape(eq, v, **kw):
    def sp(t):
      t = t.replace("...", "@")
      out = "".join(f" {c} " if (c.isalnum() or c == "@") else c for c in t)
      return out.replace("@", "...")
    lhs, rhs = eq.split("->")
    return einops.rearrange(v, sp(lhs) + " -> " + sp(rhs), **kw)

  return einshape


def import_reference():
  """Returns (tapir_model module, utils module, nets module) of the reference."""
  if not reference_available():
    raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
  for name in ("einshape", "einshape.src", "einshape.src.abstract_ops",
               "einshape.src.backend"):
    sys.modules.setdefault(name, types.ModuleType(name))

  class _Backend:  # tapnet/torch/utils.py:234 subclasses backend.Backend[Tensor]
    def __class_getitem__(cls, k):
      return cls

  sys.modules["einshape.src.backend"].Backend = _Backend
  ao = sys.modules["einshape.src.abstract_ops"]
  ao.Reshape = ao.Transpose = ao.Broadcast = object

  tree = types.ModuleType("tree")

  def map_structure(f, *s):
    a = s[0]
    if isinstance(a, (list, tuple)):
      return type(a)(map_structure(f, *[x[i] for x in s]) for i in range(len(a)))
    if isinstance(a, dict):
      return {k: map_structure(f, *[x[k] for x in s]) for k in a}
    return f(*s)

  tree.map_structure = map_structure
  sys.modules.setdefault("tree", tree)
  if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)
  from tapnet.torch import utils as tu
  tu.einshape = _einshape_adapter()
  from tapnet.torch import tapir_model as tm
  from tapnet.torch import nets as tn
  return tm, tu, tn
