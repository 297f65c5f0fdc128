#!/usr/bin/env python
"""HBM traffic per launch of the dominant kernels from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE need 3 + 2 of the 4 TCC slots: separate runs), corrected as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: both counters are in KiB; on gfx950
FETCH_SIZE reports half the bytes of wide (16 B / lane) coalesced streaming reads, so it is
doubled; WRITE_SIZE is taken as reported (uncalibrated in the guide).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/pmc_traffic.json
"""
import collections
import csv
import glob
import json
import os
import sys

KERNELS = {
    'gemm_up': ('gemm_nt_kernel<unsigned short, unsigned short, 1', 'GemmTile<2, 4, 4, 2, 2'),   # 128x128, 8 waves
    'gemm_down': ('gemm_nt_kernel<unsigned short, float, 2', 'GemmTile<2, 2, 6, 2, 2'),          # 192x64
    'mix': ('mix_stream_kernel<unsigned short, 12, false', ''),
    # round 2
    'mixer_fused': ('mixer_fused_kernel<unsigned short, 3, false', ''),
    'mixer_fused_wide': ('mixer_fused_wide_kernel<3, 2, false', ''),
    'cv_fused': ('cv_fused_kernel<unsigned short', ''),
    'patch_corr': ('patch_corr_kernel<unsigned short', ''),
    'conv3x3_c64_shortcut': ('conv_fused_kernel<', '64, 64, 3, 1, 4, 4, true'),
    'conv3x3_c64': ('conv_fused_kernel<', '64, 64, 3, 1, 4, 4, false'),
    'conv3x3_c128_shortcut': ('conv_fused_kernel<', '128, 128, 3, 1, 4, 4, true'),
    'conv3x3_c256_shortcut': ('conv_fused_kernel<', '256, 256, 3, 1, 4, 4, true'),
    'conv3x3_c256': ('conv_fused_kernel<', '256, 256, 3, 1, 4, 4, false'),
    'conv3x3_s2_64_128': ('conv_fused_kernel<', '64, 128, 3, 2'),
    'stem': ('stem_conv_kernel', ''),
    # round 3
    'xconv_256_1024_gelu': ('xconv_kernel<unsigned short, 256, true, false', ''),
    'xconv_1024_256_skip': ('xconv_kernel<unsigned short, 256, false, true', ''),
    'ln_affine': ('ln_affine_kernel<unsigned short', ''),
    # round 4
    'cv_rows': ('cv_rows_kernel<unsigned short', ''),
    'l2norm': ('l2norm_kernel<unsigned short', ''),
    'pool_cast': ('pool_cast_kernel<unsigned short', ''),
    # round 6 (the online frame: tools/r06.sh onlpmc)
    'mixer_online': ('mixer_online_kernel<unsigned short', ''),
    'mlp_small': ('mlp_small_kernel<unsigned short', ''),
    'mix_few_rows': ('mix_kernel<unsigned short', ''),
}


def rows_of(d):
  """(kernel name, counter name, value) of every dispatch: rocprofv3's CSV or rocpd (sqlite) output"""
  f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
  if f:
    for r in csv.DictReader(open(f[0])):
      yield r['Kernel_Name'], r['Counter_Name'], float(r['Counter_Value'])
    return
  import sqlite3
  db = sqlite3.connect(glob.glob(os.path.join(d, '**', '*results.db'), recursive=True)[0])
  for row in db.execute('select kernel_name, counter_name, value from counters_collection'):
    yield row[0], row[1], float(row[2])


def per_kernel(d, counter):
  acc = collections.defaultdict(list)
  for name, cname, value in rows_of(d):
    if cname != counter:
      continue
    for key, (a, b) in KERNELS.items():
      if a in name and b in name:
        acc[key].append(value)
  return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


BACKBONE = ('conv_fused_kernel', 'stem_conv_kernel', 'l2norm_kernel', 'inorm_', 'xconv_kernel', 'ln_affine_kernel')


def backbone_totals(d, counter):
  """sum of `counter` over every backbone dispatch of the run, and the number of clips (stem launches / 4 frame groups)"""
  tot, stems, by = 0.0, 0, collections.defaultdict(float)
  for name, cname, value in rows_of(d):
    if cname != counter or not any(b in name for b in BACKBONE):
      continue
    tot += value
    stems += 'stem_conv_kernel' in name
    fam = 'dual conv_0 + proj_conv' if ', false, false, true>' in name else next(b for b in BACKBONE if b in name)
    by[fam] += value
  return tot, stems, dict(by)


def main():
  fetch, nf = per_kernel(sys.argv[1], 'FETCH_SIZE')
  write, nw = per_kernel(sys.argv[2], 'WRITE_SIZE')
  out = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) on the command named in profiles/README.md, '
                   'FETCH_SIZE x2 (gfx950 wide-read correction), KiB -> bytes; averages per launch',
         'kernels': {}}
  for k in KERNELS:
    if k not in fetch:
      continue
    fb = fetch[k] * 1024 * 2
    wb = write.get(k, 0.0) * 1024
    out['kernels'][k] = dict(fetch_bytes=round(fb), write_bytes=round(wb), hbm_bytes=round(fb + wb),
                             fetch_kib_raw=round(fetch[k], 1), write_kib_raw=round(write.get(k, 0.0), 1),
                             launches=nf[k])
  # the backbone of ONE clip (all of its launches; graph off: four frame groups of 12 frames -> clips = stem launches / 4)
  ft, fs, fby = backbone_totals(sys.argv[1], 'FETCH_SIZE')
  wt, ws, wby = backbone_totals(sys.argv[2], 'WRITE_SIZE')
  if fs and ws:
    groups = int(os.environ.get('PMC_FRAME_GROUPS', '4'))
    cf, cw = fs / groups, ws / groups
    out['backbone_per_clip'] = dict(
        fetch_bytes=round(ft * 1024 * 2 / cf), write_bytes=round(wt * 1024 / cw),
        hbm_bytes=round(ft * 1024 * 2 / cf + wt * 1024 / cw), clips_fetch_pass=cf, clips_write_pass=cw,
        by_family_fetch_MB={k: round(v * 1024 * 2 / cf / 1e6, 1) for k, v in fby.items()},
        by_family_write_MB={k: round(v * 1024 / cw / 1e6, 1) for k, v in wby.items()},
        note='every backbone dispatch of the run summed (FETCH_SIZE x 2, WRITE_SIZE), divided by the clips of that pass')
  if 'gemm_up' in out['kernels']:
    out['gemm_up_hbm_bytes_per_launch'] = out['kernels']['gemm_up']['hbm_bytes']
  json.dump(out, sys.stdout, indent=1)
  print()


if __name__ == '__main__':
  main()
