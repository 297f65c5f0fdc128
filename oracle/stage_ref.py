"""TEST / MEASUREMENT INFRASTRUCTURE ONLY -- stages the reference's own CPU path next to the oracle so that it can be
TIMED ON THE GPU BOX ("the reference's CPU path timed beside it").

The reference is pure Python: its torch twin (tapnet/torch/{tapir_model,nets,utils}.py, the only reference path that
runs offline -- no jax wheel) needs no build, only to be present.  /root/reference does not exist on the GPU box, so
this recipe copies those three files (plus the two package __init__.py they are imported through) from where they lie
under /root/reference into oracle/_ref/ -- git-ignored, never committed, shipped with the push exactly like the built
.so files -- and oracle.ref_import.import_reference() falls back to that directory when /root/reference is absent.

Nothing in tapnet_amd/ may import from here; bench.py's `cpu_baseline` leg and tests are the only users.

    python -m oracle.stage_ref            (called by __graft_entry__.build() when /root/reference exists)
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DEST = os.path.join(HERE, '_ref')
SRC = os.environ.get('TAPNET_REFERENCE', '/root/reference')
FILES = ('tapnet/__init__.py', 'tapnet/torch/__init__.py', 'tapnet/torch/tapir_model.py', 'tapnet/torch/nets.py',
         'tapnet/torch/utils.py')


def stage(verbose: bool = True) -> bool:
  if not os.path.isdir(os.path.join(SRC, 'tapnet', 'torch')):
    if verbose:
      print(f'oracle/stage_ref: no reference tree at {SRC}; keeping whatever oracle/_ref holds')
    return os.path.isfile(os.path.join(DEST, 'tapnet', 'torch', 'tapir_model.py'))
  for rel in FILES:
    dst = os.path.join(DEST, rel)
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    shutil.copyfile(os.path.join(SRC, rel), dst)
  with open(os.path.join(DEST, 'README'), 'w') as f:
    f.write('Staged by oracle/stage_ref.py from the read-only reference tree: NOT part of this repository (git-ignored).\n'
            'Only bench.py\'s cpu_baseline leg and tests import it, through oracle/ref_import.py.\n')
  if verbose:
    print(f'oracle/stage_ref: staged {len(FILES)} reference files into {os.path.relpath(DEST)}')
  return True


if __name__ == '__main__':
  sys.exit(0 if stage() else 1)
