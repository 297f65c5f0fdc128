#!/usr/bin/env python
"""Kernel timeline of the LAST synchronised call in a rocprofv3 rocpd database:
    python tools/timeline.py gpurun_out/prof/x_results.db [gap_us=200000]
Calls are separated by host synchronisation, i.e. by the largest idle gaps; prints per-kernel-name
totals, the busy time and the idle time between consecutive kernels of the last call."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, start, end from kernels order by start').fetchall()
n_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# the last n_calls calls are separated by the n_calls largest gaps near the end; take the last segment
gaps = [(rows[i + 1][1] - rows[i][2], i) for i in range(len(rows) - 1)]
cut = sorted(sorted(gaps)[-(n_calls - 1):], key=lambda g: g[1])[-1][1] + 1 if n_calls > 1 else 0
seg = rows[cut:]
t0, t1 = seg[0][1], seg[-1][2]
busy = sum(e - s for _, s, e in seg)
idle = [(seg[i + 1][1] - seg[i][2]) for i in range(len(seg) - 1)]
print(f'last call: {len(seg)} kernels, span {(t1 - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, '
      f'idle {sum(max(g, 0) for g in idle) / 1e3:.1f} us (overlap {-sum(min(g, 0) for g in idle) / 1e3:.1f} us)')
import statistics
pos = [g / 1e3 for g in idle if g > 0]
print(f'gaps: median {statistics.median(pos):.2f} us, p90 {sorted(pos)[int(len(pos) * 0.9)]:.2f} us, max {max(pos):.1f} us')
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in seg:
  agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
  print(f'{t:9.1f} us {c:4d} x {t / c:8.2f}  {n[:110]}')
big = sorted(((g / 1e3, seg[i][0][:50], seg[i + 1][0][:50]) for i, g in enumerate(idle)), reverse=True)[:8]
for g, a, b in big:
  print(f'gap {g:8.1f} us between {a} -> {b}')
