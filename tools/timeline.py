"""Timeline of one steady-state clip from a rocprofv3 --kernel-trace rocpd database: per kernel start / end relative to
the clip's first dispatch, the queue it ran on, idle gaps of the whole device and the tail of every backbone stream.
usage: python tools/timeline.py gpurun_out/xxx/prof/bench_results.db [clip index from the end, default 3]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = db.execute('select name, start, end, queue_id, stream_id from kernels order by start').fetchall()
short = lambda n: n.replace('void tapir::', '').replace('tapir::', '').replace('unsigned short', 'bf16')[:64]
# a clip of the headline loop = the dispatches between two consecutive stem launches' groups; anchor on the LAST mixer
# launch of a clip: a 3-tile mixer followed (eventually) by a stem kernel or the end
mix = [i for i, r in enumerate(rows) if 'mixer_fused_kernel<unsigned short, 3' in r[0]]
# group mixers into clips of 4 consecutive launches with no stem in between
stems = [i for i, r in enumerate(rows) if 'stem_conv_kernel' in r[0]]
clips = []
for k in range(0, len(mix) - 3):
  a, b = mix[k], mix[k + 3]
  if not any(a < s < b for s in stems) and (k == 0 or any(mix[k - 1] < s < a for s in stems)):
    first = max([s for s in stems if s < a][-4:][0:1] or [a])
    clips.append((first, b))
if not clips:
  sys.exit('no headline clip found')
first, last = clips[-back] if len(clips) >= back else clips[-1]
# the clip starts with whatever precedes the first stem of its group (copy kernels): walk back over non-tapir kernels
i0 = first
while i0 > 0 and 'tapir' not in rows[i0 - 1][0] and rows[first][1] - rows[i0 - 1][2] < 100000:
  i0 -= 1
t0 = rows[i0][1]
sel = rows[i0:last + 1]
print(f'clip: {len(sel)} dispatches, {(rows[last][2] - t0) / 1e3:.1f} us from first start to last end')
queues = sorted({r[3] for r in sel})
for q in queues:
  qs = [r for r in sel if r[3] == q]
  busy = sum(r[2] - r[1] for r in qs)
  print(f'queue {q}: {len(qs)} dispatches, busy {busy / 1e3:.1f} us, from {(qs[0][1] - t0) / 1e3:.1f} to {(qs[-1][2] - t0) / 1e3:.1f} us')
# device-level idle gaps (no kernel of the clip running)
ev = sorted((r[1], r[2]) for r in sel)
cur_end, idle, gaps = ev[0][1], 0, []
for s, e in ev[1:]:
  if s > cur_end:
    idle += s - cur_end
    gaps.append(((cur_end - t0) / 1e3, (s - cur_end) / 1e3))
  cur_end = max(cur_end, e)
print(f'device idle inside the clip: {idle / 1e3:.1f} us in {len(gaps)} gaps; largest:', sorted(gaps, key=lambda g: -g[1])[:8])
if '-v' in sys.argv:
  for r in sel:
    print(f'{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  q{r[3]}  {short(r[0])}')
else:   # the serial part after the backbone joins
  main_q = max(queues, key=lambda q: sum(1 for r in sel if r[3] == q and 'mixer' in r[0]))
  for r in sel:
    if r[3] == main_q and ('conv_fused' not in r[0]):
      print(f'{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  q{r[3]}  {short(r[0])}')
