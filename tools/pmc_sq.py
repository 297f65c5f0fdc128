#!/usr/bin/env python
"""Per-kernel SQ counters from one rocprofv3 --pmc pass (see tools/r05.sh prof / profiles/README.md):
wave cycles, wait / active shares, MFMA busy cycles, LDS bank conflicts, averaged per launch.

    python tools/pmc_sq.py gpurun_out/<dir>/pmc_sq > profiles/rNN_pmc_sq.txt
"""
import collections
import sys

from pmc_traffic import KERNELS, rows_of

COUNTERS = ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY',
            'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'GRBM_GUI_ACTIVE']


def main():
  acc = collections.defaultdict(lambda: collections.defaultdict(list))
  for name, cname, value in rows_of(sys.argv[1]):
    for key, (a, b) in KERNELS.items():
      if a in name and b in name:
        acc[key][cname].append(value)
  print('rocprofv3 --pmc ' + ' '.join(COUNTERS))
  print('per launch (mean); SQ_* are summed over the chip; wait / active as shares of SQ_WAVE_CYCLES; '
        'mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8): share of the launch the matrix pipes '
        'are busy (GRBM_GUI_ACTIVE is summed over the 8 XCDs)')
  print(f'{"kernel":26s} {"launches":>8s} {"gui_active":>11s} {"wave_cyc":>10s} {"wait_any":>8s} {"wait_inst":>9s} {"active":>7s} '
        f'{"mfma_busy":>10s} {"mfma":>6s} {"lds_conflict":>12s} {"lds_active":>10s} {"conflict/active":>15s}')
  for key in KERNELS:
    c = acc.get(key)
    if not c or 'SQ_WAVE_CYCLES' not in c:
      continue
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m['SQ_WAVE_CYCLES']
    gui = m.get('GRBM_GUI_ACTIVE', float('nan'))
    mf = m.get('SQ_VALU_MFMA_BUSY_CYCLES', float('nan'))
    la = m.get('SQ_LDS_IDX_ACTIVE', float('nan'))
    lc = m.get('SQ_LDS_BANK_CONFLICT', float('nan'))
    print(f'{key:26s} {len(c["SQ_WAVE_CYCLES"]):8d} {gui:11.0f} {wc:10.3g} {m.get("SQ_WAIT_ANY", 0) / wc:8.2f} '
          f'{m.get("SQ_WAIT_INST_ANY", 0) / wc:9.2f} {m.get("SQ_ACTIVE_INST_ANY", 0) / wc:7.2f} {mf:10.3g} '
          f'{mf / (1024 * gui / 8):6.2f} {lc:12.3g} {la:10.3g} {lc / la if la else float("nan"):15.3f}')


if __name__ == '__main__':
  sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
  main()
