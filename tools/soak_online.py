"""Soak of the online tracker (persistent mixer launch): many replayed frames, the error word and a finite check every 1000
frames, and the last frames against a fresh eager session."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tapnet_amd import online, synthetic, tapir_model

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
S, Q = 256, 256
w = synthetic.make_weights(0, pyramid_level=1, extra_convs=True)
m = tapir_model.TAPIR(pyramid_level=1, extra_convs=True, use_causal_conv=True, weights=w, dtype='bfloat16', device='cuda:0')
video = torch.as_tensor(synthetic.make_video(1, 8, S, S)).cuda()
qp = torch.as_tensor(synthetic.make_queries(2, Q, 1, S, S)).cuda()
trk = online.OnlineTracker(m, Q, (S, S), use_graph=True)
trk.init(video[:, :1], qp)
t0 = time.perf_counter()
bad = 0
for t in range(frames):
  out = trk.step(video[:, t % 8:t % 8 + 1])
  if t % 1000 == 999:
    bad += int(not bool(torch.isfinite(out['tracks']).all()))
    trk.check()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / frames * 1e3
trk.check()
last = out['tracks'].clone()
ref = online.OnlineTracker(m, Q, (S, S), use_graph=False)
ref.init(video[:, :1], qp)
for t in range(min(frames, 64)):
  r = ref.step(video[:, t % 8:t % 8 + 1])
trk2 = online.OnlineTracker(m, Q, (S, S), use_graph=True)
trk2.init(video[:, :1], qp)
for t in range(min(frames, 64)):
  o2 = trk2.step(video[:, t % 8:t % 8 + 1])
print(f'{frames} replayed frames, {ms:.3f} ms per frame (with a check every 1000), non-finite checkpoints: {bad}, error word: none; '
      f'frame 64 replay vs eager: max |diff| {float((o2["tracks"] - r["tracks"]).abs().max()):.3g} px, last frame finite: {bool(torch.isfinite(last).all())}')
