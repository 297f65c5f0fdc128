"""Is the bf16 engine run-to-run deterministic on a tiny shape, with grids given, staged and plain?  (GPU probe.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tapnet_amd import synthetic, tapir_model, distributed as tdist

dev = torch.device('cuda:0')
S, T, Q = 64, 9, 5
w = synthetic.make_weights(17, pyramid_level=1, extra_convs=False)
m = tapir_model.TAPIR(pyramid_level=1, weights=w, device=dev, initial_resolution=(S, S), dtype='bfloat16')
video = torch.as_tensor(synthetic.make_video(3, T, S, S), device=dev)
qp = torch.as_tensor(synthetic.make_queries(4, Q, T, S, S), device=dev)
keys = ('tracks', 'occlusion', 'expected_dist')
dmax = lambda a, b: max(float((a[k] - b[k]).abs().max()) for k in keys)
r = [m(video, False, qp) for _ in range(3)]
print('video -> tracks, 3 runs:', dmax(r[0], r[1]), dmax(r[1], r[2]))
# grids as the sharded call hands them over: f32 = the bf16 copies converted back, staged copies next to them
m._staged = []
fg = m.get_feature_grids(video, _borrow=True)
staged, m._staged = list(m._staged), []
own = [(f.clone(), op.clone(), None if tl is None else tl.clone()) for f, op, tl in staged]
lo16, hi16 = own[0][1], own[1][1]
lo32 = lo16.float().reshape(fg.lowres[0].shape); hi32 = hi16.float().reshape(fg.hires[0].shape)
n = len(fg.lowres)
sfg = tapir_model.StagedFeatureGrids((lo32,) * n, (hi32,) * n, fg.resolutions)
sfg.staged = [(lo32, lo16, own[0][2]), (hi32, hi16, None)]
plain = tapir_model.FeatureGrids((lo32,) * n, (hi32,) * n, fg.resolutions)
sh = tdist.ShapeOnly(video.shape)
a = [m(sh, False, qp, feature_grids=sfg) for _ in range(3)]
b = [m(sh, False, qp, feature_grids=plain) for _ in range(3)]
print('staged x3:', dmax(a[0], a[1]), dmax(a[1], a[2]), '| plain x3:', dmax(b[0], b[1]), dmax(b[1], b[2]), '| staged vs plain:', dmax(a[0], b[0]), dmax(a[2], b[2]))
# the tile-order copy against the row-major one
tl = own[0][2].reshape(T, -1, 32, 16, 8)
rm = lo16.reshape(T, -1, 16, 32, 8).permute(0, 1, 3, 2, 4)
print('tile-order copy == row-major copy re-tiled:', bool(torch.equal(tl, rm.contiguous())))
for Q2 in (5, 10, 64):
  qp2 = torch.as_tensor(synthetic.make_queries(4, Q2, T, S, S), device=dev)
  x = [m(sh, False, qp2, feature_grids=plain) for _ in range(2)]
  y = m(sh, False, qp2[:, :Q2 // 2 + 1], feature_grids=plain)
  print('Q', Q2, 'plain twice:', dmax(x[0], x[1]), '| first half alone vs inside the batch:',
        max(float((x[0][k][:, :Q2 // 2 + 1] - y[k]).abs().max()) for k in keys))
