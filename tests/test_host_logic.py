"""CPU tests of the host layer: C-ABI symbols, model_utils mirror, backbone
restatement vs the reference-generated fixture, checkpoint name conversion."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tapnet_amd import _ffi, backbone, model_utils, synthetic, weights
from tests.golden_util import GOLDEN_DIR


def test_abi_symbols_exported():
  """the gfx950 library loads on a CPU-only host and exports every declared symbol."""
  if not os.path.exists(_ffi.LIB_PATH):
    import __graft_entry__
    __graft_entry__.build()
  lib = _ffi.declare_prototypes(ctypes.CDLL(_ffi.LIB_PATH))
  assert b'gfx950' in lib.tapir_version()
  hdr = open(os.path.join(os.path.dirname(GOLDEN_DIR), '..', 'include', 'tapir_hip.h')).read()
  for name in _ffi.PROTOTYPES:
    assert name + '(' in hdr, name


def test_product_refuses_cpu():
  from tapnet_amd import tapir_model
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  with pytest.raises(RuntimeError):
    tapir_model.TAPIR()


def test_generate_default_resolutions():
  assert model_utils.generate_default_resolutions((256, 256), (256, 256)) == [(256, 256)]
  assert model_utils.generate_default_resolutions((512, 512), (256, 256)) == [(256, 256), (512, 512)]
  assert model_utils.generate_default_resolutions((480, 640), (256, 256)) == \
      [(256, 256), (344, 400), (480, 640)]


def test_convert_grid_coordinates_errors():
  c = np.ones((2, 3), np.float32)
  with pytest.raises(ValueError):
    model_utils.convert_grid_coordinates(c, (1, 2, 3), (2, 2, 3), 'tyx')
  with pytest.raises(ValueError):
    model_utils.convert_grid_coordinates(c, (1, 2), (2, 2), 'abc')
  out = model_utils.convert_grid_coordinates(c, (4, 8, 8), (4, 2, 4), 'tyx')
  np.testing.assert_allclose(out, [[1, 0.25, 0.5]] * 2)


def test_postprocess_occlusions():
  v = model_utils.postprocess_occlusions(np.array([-5.0, 5.0, -5.0]), np.array([-5.0, -5.0, 5.0]))
  assert v.tolist() == [True, False, False]


@pytest.mark.parametrize('tag,extra', [('tapir', False), ('boots', True)])
def test_backbone_matches_reference_fixture(tag, extra):
  """R7 restatement (torch ops, CPU here) vs the reference's feature grids."""
  g = np.load(os.path.join(GOLDEN_DIR, 'backbone.npz'))
  w = synthetic.make_weights(21, 1, extra)
  from oracle import backbone_torch
  bb = backbone_torch.TorchBackbone(w, extra)
  low, hi = bb.features(torch.as_tensor(g['video']).reshape(-1, 64, 64, 3))
  np.testing.assert_allclose(low.numpy(), g[f'{tag}_lowres'][0], atol=2e-5)
  np.testing.assert_allclose(hi.numpy(), g[f'{tag}_hires'][0], atol=2e-5)


def test_haiku_name_conversion_shapes():
  """round trip: torch-named weights -> fake Haiku params -> torch names."""
  w = synthetic.make_weights(1, 1, True, num_mixer_blocks=2)
  hk = {}
  def conv(name, t): hk[name] = {'w': np.transpose(w[t + '.weight'], (2, 3, 1, 0))}
  root = 'tapir/~/'
  for t, h in {'hid1': 'cost_volume_regression_1', 'hid2': 'cost_volume_regression_2',
               'hid3': 'cost_volume_occlusion_1'}.items():
    conv(root + h, 'torch_cost_volume_track_mods.' + t)
    hk[root + h]['b'] = w[f'torch_cost_volume_track_mods.{t}.bias']
  for t, h in {'hid4': 'cost_volume_occlusion_2', 'occ_out': 'occlusion_out'}.items():
    hk[root + h] = {'w': w[f'torch_cost_volume_track_mods.{t}.weight'].T,
                    'b': w[f'torch_cost_volume_track_mods.{t}.bias']}
  mx = root + 'pips_mlp_mixer/'
  for t in ('linear', 'linear_1'):
    hk[mx + t] = {'w': w[f'torch_pips_mixer.{t}.weight'].T, 'b': w[f'torch_pips_mixer.{t}.bias']}
  hk[mx + 'layer_norm'] = {'scale': w['torch_pips_mixer.layer_norm.weight']}
  for i in range(2):
    blk = mx + ('block' if i == 0 else f'block_{i}') + '/'
    p = f'torch_pips_mixer.blocks.{i}.'
    hk[blk + 'layer_norm'] = {'scale': w[p + 'layer_norm.weight']}
    hk[blk + 'layer_norm_1'] = {'scale': w[p + 'layer_norm_1.weight']}
    for t in ('mlp1_up', 'mlp1_up_1'):
      hk[blk + t] = {'w': np.transpose(w[p + t + '.weight'], (2, 1, 0)), 'b': w[p + t + '.bias']}
    for t in ('mlp2_up', 'mlp2_down'):
      hk[blk + t] = {'w': w[p + f'conv_channels_mixer.{t}.weight'].T,
                     'b': w[p + f'conv_channels_mixer.{t}.bias']}
  assert weights.is_haiku_params(hk)
  back = weights.to_torch_names(hk)
  for k, v in back.items():
    np.testing.assert_array_equal(v, w[k])
  assert 'torch_pips_mixer.blocks.1.mlp1_up.weight' in back


def test_tapnet_haiku_param_conversion():
  """tapnet_amd.tapnet_model.from_haiku_params: hk.Conv3D kernels [1,3,3,in,out] / hk.Linear [in,out]
  (tapnet/models/tapnet_model.py:64-107) -> the torch layout of the TAPIR head the engine loads."""
  import numpy as np
  from tapnet_amd import tapnet_model
  rng = np.random.default_rng(0)
  shapes = {'cost_volume_regression_1': (1, 3, 3, 1, 16), 'cost_volume_regression_2': (1, 3, 3, 16, 1),
            'cost_volume_occlusion_1': (1, 3, 3, 16, 32), 'cost_volume_occlusion_2': (32, 16),
            'occlusion_out': (16, 1)}
  params = {f'tap_net/{k}': {'w': rng.standard_normal(s).astype(np.float32),
                             'b': rng.standard_normal(s[-1]).astype(np.float32)} for k, s in shapes.items()}
  flat = tapnet_model.from_haiku_params(params)
  assert flat['tapnet_cost_volume_track_mods.hid3.weight'].shape == (32, 16, 3, 3)
  assert flat['tapnet_cost_volume_track_mods.occ_out.weight'].shape == (1, 16)
  w = params['tap_net/cost_volume_occlusion_1']['w']
  assert flat['tapnet_cost_volume_track_mods.hid3.weight'][5, 7, 2, 1] == w[0, 2, 1, 7, 5]
  assert flat['tapnet_cost_volume_track_mods.hid4.weight'][3, 9] == params['tap_net/cost_volume_occlusion_2']['w'][9, 3]


def test_bench_self_launch_command(monkeypatch):
  """`python bench.py --gpus N` (N > 1, no launcher around it) re-executes itself through
  torch.distributed.run with one process per GPU and a 127.0.0.1 rendezvous, passing its flags on."""
  import importlib
  import subprocess
  import sys
  bench = importlib.import_module('bench')
  seen = {}

  def fake_call(cmd, env=None):
    seen['cmd'], seen['env'] = cmd, env
    return 0

  monkeypatch.setattr(subprocess, 'call', fake_call)
  monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3'])
  args = bench.parse()
  assert bench.self_launch(args) == 0
  cmd = seen['cmd']
  assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=4' in cmd
  assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
  assert cmd[-4:] == ['--gpus', '4', '--steps', '3'] and cmd[-5].endswith('bench.py')
  assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_jax_style_antialiased_resize_matches_the_restatement():
  """TAPIR(jax_antialias_resize=True): the JAX model's jax.image.resize(method='bilinear') anti-aliases when
  DOWN-sampling (tapir_model.py:670), the torch twin does not (tapnet/torch/utils.py:39).  The product's
  resize_bilinear(antialias=True) against the numpy restatement of jax/_src/image/scale.py's published algorithm
  (oracle/jax_resize.py; JAX itself cannot run offline): 512 -> 256 (BASELINE configs[4]'s first level), a non-integer
  factor, and up-sampling, where both agree with plain bilinear interpolation."""
  import torch
  from oracle import jax_resize
  from tapnet_amd.backbone import resize_bilinear
  rng = np.random.default_rng(0)
  v = rng.uniform(-1, 1, (1, 2, 64, 48, 3)).astype(np.float32)
  for res in ((32, 24), (40, 40), (24, 20)):                     # down-sampling: the antialiased kernel
    got = resize_bilinear(torch.as_tensor(v), res, antialias=True).numpy()
    np.testing.assert_allclose(got, jax_resize.resize_bilinear(v, res), atol=2e-5)   # (measured 6e-8 .. 4e-6: float32 weights)
    plain = resize_bilinear(torch.as_tensor(v), res, antialias=False).numpy()
    assert np.abs(plain - got).max() > 0.05                      # ... which the torch twin's resize is not
  up = (96, 80)                                                  # up-sampling: identical to plain bilinear
  got = resize_bilinear(torch.as_tensor(v), up, antialias=True).numpy()
  np.testing.assert_allclose(got, jax_resize.resize_bilinear(v, up), atol=2e-5)
  np.testing.assert_allclose(got, resize_bilinear(torch.as_tensor(v), up, antialias=False).numpy(), atol=1e-5)
  # the 2x case has the closed form [1, 3, 3, 1] / 8 away from the borders
  w = jax_resize.compute_weight_mat(8, 4)
  np.testing.assert_allclose(w[1:5, 1], [0.125, 0.375, 0.375, 0.125], atol=1e-12)


def test_resolution_answers_both_conventions():
  """FeatureGrids / QueryFeatures.resolutions entries: (H, W) tuples as in the torch twin
  (tapnet/torch/tapir_model.py:45) that also answer `.shape[:2]` like the JAX model's zero-size shape carriers
  (tapnet/models/tapir_model.py:259-265, 724)."""
  from tapnet_amd.tapir_model import Resolution, _res_hw
  r = Resolution((64, 96))
  assert r == (64, 96) and r[0] == 64 and tuple(r) == (64, 96)
  assert r.shape == (64, 96, 0) and r.shape[:2] == (64, 96)
  assert _res_hw(r) == (64, 96) and _res_hw((64, 96)) == (64, 96)
  import numpy as np
  assert _res_hw(np.zeros((64, 96, 0), np.float32)) == (64, 96)       # a JAX-style carrier passed in


def test_resize_to_the_same_size_is_the_identity():
  """resize_bilinear returns its input when the target size is the video's own (the reference's first level always
  resizes, tapir_model.py:667-670): pin that what it skips -- F.interpolate(bilinear, align_corners=False) of the
  reference's torch twin (tapnet/torch/utils.py:26-42), with and without anti-aliasing, and the restatement of
  jax.image.resize -- is the identity bit for bit, so the shortcut changes no output."""
  import torch
  import torch.nn.functional as F
  from tapnet_amd.backbone import resize_bilinear
  from oracle import jax_resize
  rng = np.random.default_rng(11)
  v = rng.uniform(-1, 1, (1, 3, 24, 40, 3)).astype(np.float32)
  v[0, 0, 0, 0, 0] = 0.0
  t = torch.as_tensor(v)
  x = t.permute(0, 1, 4, 2, 3).reshape(1, 9, 24, 40)
  for aa in (False, True):
    y = F.interpolate(x, size=(24, 40), mode='bilinear', align_corners=False, antialias=aa)
    assert torch.equal(y, x), aa
    out = resize_bilinear(t, (24, 40), antialias=aa)
    assert out.shape == t.shape and torch.equal(out, t)
  np.testing.assert_array_equal(jax_resize.resize_bilinear(v, (24, 40)), v)


def test_bench_roofline_all_prices_every_kernel_class():
  """bench.py's `roofline_all` (round 5): one entry per kernel class with the algorithmic work DESIGN.md 3 states --
  the mixer's three GEMM families, the cost volume's composite floor (operand-type einsum + hid3, exact-f32 hid1 / hid2),
  unique bytes for the memory-bound classes, and the 1x1 projections booked where the dual launches run them."""
  import importlib.util
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  T, Q, S = 48, 256, 256
  prof = {'mixer_fused': (4 * 0.700, 4), 'cv_heads': (0.172, 1), 'patch_corr': (4 * 0.035, 4)}
  bbprof = {'stem': (0.077, 1), 'conv3x3_c64': (4 * 0.112, 4), 'conv3x3_c128': (3 * 0.088, 3), 'conv3x3_c256': (7 * 0.086, 7),
            'conv_other': (0.196, 2), 'l2norm': (0.043, 2)}
  ra = bench.roofline_all(prof, bbprof, T, Q, S, 2, 'bfloat16', 0, dual=True)
  assert set(ra) == {'mixer_fused', 'cv_rows', 'patch_corr', 'conv3x3_c64', 'conv3x3_c128', 'conv3x3_c256', 'conv_other',
                     'stem', 'l2norm'}
  R = Q * T
  assert ra['mixer_fused']['flops'] == 2.0 * R * (486 * 512 + 12 * 2 * 512 * 2048 + 512 * 388)      # 629.5 GFLOP
  np.testing.assert_allclose(ra['mixer_fused']['frac'], 629.47e9 / 700e-6 / 2.5e15, rtol=1e-3)
  # the track-resident form's second bound: the packed weight set once per workgroup through the L2 -> CU path
  ws = ra['mixer_fused']['weight_stream']
  assert ws['bytes_per_workgroup'] == (512 * 512 + 12 * 2 * 512 * 2048 + 512 * 512) * 2                 # 50.9 MB (486 -> 512)
  np.testing.assert_allclose(ws['frac'], ws['bytes_per_workgroup'] / 700e-6 / 1e9 / 128.4, rtol=1e-3)   # 0.566
  assert 'weight_stream' not in bench.roofline_all(prof, bbprof, T, 1024, S, 2, 'bfloat16', 0)['mixer_fused']   # (wide form: shared)
  cells = R * 32 * 32
  np.testing.assert_allclose(ra['cv_rows']['flops'], 2.0 * cells * 256 + 2.0 * 144 * 32 * cells / 4 + 2.0 * 2 * 144 * cells)
  np.testing.assert_allclose(ra['cv_rows']['composite_floor_us'], 60.2, atol=0.1)                      # 35.4 GF bf16 + 7.2 GF f32
  base = 2.0 * T * 128 * 128 * 64 * 64 * 9
  proj = 2.0 * T * 128 * 128 * 64 * 64
  assert ra['conv3x3_c64']['flops'] == base + proj / 4 and ra['conv3x3_c128']['flops'] == 2.0 * T * 64 * 64 * 128 * 128 * 9
  two = bench.roofline_all(prof, bbprof, T, Q, S, 2, 'bfloat16', 0, dual=False)
  assert two['conv3x3_c64']['flops'] == base and two['conv_other']['flops'] > ra['conv_other']['flops']
  for k, v in ra.items():
    assert v['bound'] in ('mfma', 'hbm') and 0 < v['frac'] < 1.2 and v['peak'] in (2500.0, 8000.0), (k, v)
  assert ra['stem']['bytes'] == T * S * S * 3 * 4 + T * 128 * 128 * 64 * 2


def test_the_built_library_has_no_high_half_selects_on_packed_f32_arithmetic():
  """csrc/check_packed_forms.py on the shipped library: no `v_pk_{fma,mul,add}_f32 ... op_sel:[` in the gfx950 code object.
  On MI355X that form loses its low result in lanes 48-63 next to another wave's MFMAs (tools/micro/run_cotenant_repro.py,
  profiles/r06_cotenant_repro.txt); the build drops hipcc's SLP pass, the only source of it."""
  import importlib.util
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  lib = os.path.join(root, 'tapnet_amd', 'csrc', 'libtapir_hip.so')
  if not os.path.exists(lib):
    pytest.skip('library not built')
  spec = importlib.util.spec_from_file_location('check_packed_forms', os.path.join(root, 'tapnet_amd', 'csrc', 'check_packed_forms.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  res = mod.scan(lib)
  if res is None:
    pytest.skip('llvm-objdump not installed')
  packed, bad = res
  assert packed > 10000, packed          # (the hand-packed kernels are there)
  assert not bad, bad
