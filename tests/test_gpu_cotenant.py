"""`-m gpu`: the token-mixing kernel of the separate-launch mixer stays bit-identical while ANOTHER stream of the process
issues MFMAs back to back on the same SIMDs.

Why this is a test: on MI355X a packed FMA whose low result reads the high half of a source (`v_pk_fma_f32 ... op_sel:[0,1,0]`)
loses that result in lanes 48-63 next to another wave's MFMAs (stand-alone reproducer tools/micro/run_cotenant_repro.py,
profiles/r06_cotenant_repro.txt).  hipcc's SLP vectoriser used to write that form into mix_kernel (mixer.hpp; the reference's
arithmetic, tapir_model.py:39-89, is deterministic): built that way this test fails on practically every launch.  The library
is built without the pass (csrc/build.sh) and csrc/check_packed_forms.py guards the code object; this is the behavioural half
of that guard."""
import ctypes
import os
import subprocess

import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MICRO = os.path.join(ROOT, 'tools', 'micro')


def _aggressors():
  so = os.path.join(MICRO, 'libcotenant.so')
  if not os.path.exists(so):   # (__graft_entry__.build() compiles it; a bare checkout builds it here)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O2', '-shared', '-fPIC',
                           os.path.join(MICRO, 'cotenant_aggressors.hip'), '-o', so])
  lib = ctypes.CDLL(so)
  lib.aggr_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
  return lib


@pytest.mark.parametrize('kind', [3, 4])   # MFMA C/D in AGPRs / in VGPRs
def test_token_mixing_is_bit_stable_next_to_an_mfma_stream(kind):
  from tapnet_amd import synthetic, tapir_model
  dev = torch.device('cuda', 0)
  aggr = _aggressors()
  w = synthetic.make_weights(17, pyramid_level=1, extra_convs=False, backbone=False)
  m = tapir_model.TAPIR(pyramid_level=1, weights=w, device=dev, initial_resolution=(64, 64), dtype='bfloat16')
  lib, ctx = m._lib, m._ctx
  N, T = 5, 9                                # (T < 12: the general kernel, two time chunks per track)
  x = torch.randn(N, T, 512, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
  xb = torch.empty(N, T, 512, device=dev)
  xn = torch.empty(N, T, 512, device=dev, dtype=torch.bfloat16)
  buf = (torch.randn(1 << 20, device=dev) * 0.1).contiguous()
  side = torch.cuda.Stream(dev)
  s0, s1 = m._stream(), ctypes.c_void_p(side.cuda_stream)
  call = lambda: lib.tapir_debug_mix(ctx, 0, x.data_ptr(), xb.data_ptr(), xn.data_ptr(), N, T, 0, s0)
  assert call() == 0
  torch.cuda.synchronize()
  ref, ref_n = xb.clone(), xn.clone()
  bad = 0
  for r in range(1500):
    if r % 4 == 0:
      for _ in range(8):
        assert aggr.aggr_launch(kind, 64, 400, buf.data_ptr(), s1) == 0
    xb.fill_(float('nan'))
    assert call() == 0
    if not (torch.equal(xb, ref) and torch.equal(xn, ref_n)):
      bad += 1
  torch.cuda.synchronize()
  assert bad == 0, f'{bad} of 1500 launches of mix_kernel differ from the first next to MFMA kernel {kind} on a second stream'
