"""Where one online (causal, per-frame) step goes, launch by launch: from a rocprofv3 --kernel-trace rocpd database of
`tools/bench_online.py --eager-only` (or of the hipGraph replay), one steady-state frame = the dispatches between two
consecutive `stem_conv_kernel` launches (one per frame's backbone; `iter0_kernel`, the marker of rounds 5-6, no longer exists in
the frame: its copies are written by the cost-volume kernel).  Prints per kernel class: launches, busy time,
and the idle time in FRONT of its launches (device idle: no kernel of the frame running), then the frame's totals.

    python tools/online_timeline.py gpurun_out/xxx/prof/online_results.db [frame index from the end, default 3]
"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = db.execute('select name, start, end from kernels order by start').fetchall()


def short(n):
  n = n.replace('void tapir::', '').replace('tapir::', '').replace('(anonymous namespace)::', '').replace('unsigned short', 'bf16')
  for cut in ('(', ):
    if cut in n and not n.startswith('void at'):
      n = n[:n.index(cut)]
  if n.startswith('void at::native::'):
    n = 'torch:' + n[len('void at::native::'):].split('<')[0]
  return n[:72]


marks = [i for i, r in enumerate(rows) if 'iter0_kernel' in r[0]]
if len(marks) < 3:
  marks = [i for i, r in enumerate(rows) if 'stem_conv_kernel' in r[0]]
if len(marks) < back + 2:
  sys.exit('not enough frames in the trace')
a, b = marks[-back - 1], marks[-back]
sel = rows[a:b]
t0, t1 = sel[0][1], rows[b][1]
per = collections.OrderedDict()
cur_end = sel[0][1]
busy_total = idle_total = 0
for name, s, e in sel:
  k = short(name)
  ent = per.setdefault(k, [0, 0, 0])
  gap = max(0, s - cur_end)
  ent[0] += 1
  ent[1] += e - s
  ent[2] += gap
  idle_total += gap
  cur_end = max(cur_end, e)
covered = 0
ev = sorted((s, e) for _, s, e in sel)
cs, ce = ev[0]
for s, e in ev[1:]:
  if s > ce:
    covered += ce - cs
    cs, ce = s, e
  else:
    ce = max(ce, e)
covered += ce - cs
idle_total += max(0, t1 - cur_end)
print(f'frame: {len(sel)} dispatches, {(t1 - t0) / 1e3:.1f} us from this frame\'s first launch to the next frame\'s; '
      f'device busy {covered / 1e3:.1f} us, idle {idle_total / 1e3:.1f} us ({100.0 * idle_total / (t1 - t0):.0f} %)')
print(f'{"kernel":72s} {"n":>4s} {"busy us":>9s} {"avg us":>8s} {"idle before, us":>16s}')
for k, (n, busy, gap) in sorted(per.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
  print(f'{k:72s} {n:4d} {busy / 1e3:9.1f} {busy / n / 1e3:8.2f} {gap / 1e3:16.1f}')
print(f'sum of kernel durations {sum(v[1] for v in per.values()) / 1e3:.1f} us; mean gap in front of a launch '
      f'{idle_total / max(1, len(sel)) / 1e3:.2f} us')
