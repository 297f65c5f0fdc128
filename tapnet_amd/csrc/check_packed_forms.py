"""Build guard (csrc/build.sh, tests/test_host_logic.py): the gfx950 code object of a built library must not contain packed-f32
arithmetic whose LOW result reads the HIGH half of a source (`v_pk_{fma,mul,add}_f32 ... op_sel:[..1..]`): on MI355X that form
loses its low result in lanes 48-63 next to MFMA traffic of another wave (tools/micro/run_cotenant_repro.py,
profiles/r06_cotenant_repro.txt).  Exit code 0 and a one-line summary, or 1 and the offending kernels.
    python check_packed_forms.py libtapir_hip.so
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'
BAD = re.compile(r'\bv_pk_(fma|mul|add)_f32\b.*\bop_sel:\[')


def scan(lib):
  """-> (packed instructions, {kernel: offending instructions}); None if the tools are missing"""
  if not os.path.exists(OBJDUMP):
    return None
  tmp = tempfile.mkdtemp(prefix='tapir_scan_')
  try:
    so = os.path.join(tmp, 'lib.so')
    shutil.copy(lib, so)
    subprocess.run([OBJDUMP, '--offloading', so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
    cos = [f for f in os.listdir(tmp) if 'amdgcn' in f]
    if not cos:
      raise RuntimeError('no gfx950 code object in ' + lib)
    dis = subprocess.run([OBJDUMP, '-d', '--mcpu=gfx950', os.path.join(tmp, cos[0])], check=True, capture_output=True, text=True).stdout
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
  kernel, packed, bad = '?', 0, collections.Counter()
  for line in dis.split('\n'):
    if line.endswith('>:'):
      kernel = line.split('<', 1)[1][:-2]
    elif 'v_pk_' in line:
      packed += 1
      if BAD.search(line):
        bad[kernel] += 1
  return packed, dict(bad)


if __name__ == '__main__':
  res = scan(sys.argv[1])
  if res is None:
    print('check_packed_forms: llvm-objdump not found, nothing checked')
    sys.exit(0)
  packed, bad = res
  if bad:
    print(f'check_packed_forms: {sum(bad.values())} packed-f32 instructions select a high half for a low result (op_sel) in {len(bad)} kernels:')
    for k, v in sorted(bad.items(), key=lambda kv: -kv[1])[:20]:
      print(f'  {v:4d}  {k}')
    sys.exit(1)
  print(f'check_packed_forms: {packed} packed instructions, none with an op_sel modifier on packed-f32 arithmetic')
