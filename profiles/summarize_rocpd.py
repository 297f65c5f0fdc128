"""Dumps the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.
usage: python profiles/summarize_rocpd.py gpurun_out/prof/xxx_results.db > profiles/rNN_kernel_stats.csv"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cur = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels')
print('kernel,calls,total_us,avg_us,percent')
for name, calls, tot, avg, pct in cur.fetchall():
  print(f'"{name[:140]}",{calls},{tot:.1f},{avg:.2f},{pct:.2f}')   # the view reports microseconds
