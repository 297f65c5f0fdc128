"""Which fork / join shapes does hipGraph stream capture of this ROCm accept?  (tools; the backbone's branch streams)
Each case runs in its own process: a bad shape takes the process down in hipGraphInstantiate."""
import subprocess, sys

CASES = {
  'one_level': "L.wait_stream(o)\nwith torch.cuda.stream(L): a.add_(1)\nb.add_(1)\no.wait_stream(L)",
  'two_level_join_then_kernel': "L.wait_stream(o)\nwith torch.cuda.stream(L):\n  a.add_(1)\n  B.wait_stream(L)\n  with torch.cuda.stream(B): c.add_(1)\n  a.add_(1)\n  L.wait_stream(B)\n  a.add_(1)\nb.add_(1)\no.wait_stream(L)",
  'two_level_join_last': "L.wait_stream(o)\nwith torch.cuda.stream(L):\n  a.add_(1)\n  B.wait_stream(L)\n  with torch.cuda.stream(B): c.add_(1)\n  a.add_(1)\n  L.wait_stream(B)\nb.add_(1)\no.wait_stream(L)",
  'branch_joins_origin': "L.wait_stream(o)\nwith torch.cuda.stream(L):\n  a.add_(1)\n  B.wait_stream(L)\n  with torch.cuda.stream(B): c.add_(1)\n  a.add_(1)\nb.add_(1)\no.wait_stream(L)\no.wait_stream(B)",
  'cross_deps_between_first_level_forks': "L.wait_stream(o)\nB.wait_stream(o)\nwith torch.cuda.stream(L):\n  a.add_(1)\n  B.wait_stream(L)\n  with torch.cuda.stream(B): c.add_(1)\n  a.add_(1)\n  L.wait_stream(B)\n  a.add_(1)\nb.add_(1)\no.wait_stream(L)\no.wait_stream(B)",
  'cross_deps_kernel_first_on_branch': "L.wait_stream(o)\nB.wait_stream(o)\nwith torch.cuda.stream(B): c.add_(1)\nwith torch.cuda.stream(L):\n  a.add_(1)\n  B.wait_stream(L)\n  with torch.cuda.stream(B): c.add_(1)\n  a.add_(1)\n  L.wait_stream(B)\n  a.add_(1)\nb.add_(1)\no.wait_stream(L)\no.wait_stream(B)",
  'origin_forks_branch_twice': "B.wait_stream(o)\nwith torch.cuda.stream(B): c.add_(1)\nb.add_(1)\no.wait_stream(B)\nb.add_(1)\nB.wait_stream(o)\nwith torch.cuda.stream(B): c.add_(1)\nb.add_(1)\no.wait_stream(B)\nb.add_(1)",
}
TEMPLATE = '''
import torch
a = torch.zeros(1 << 20, device='cuda'); b = a.clone(); c = a.clone()
L = torch.cuda.Stream(); B = torch.cuda.Stream()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
  o = torch.cuda.current_stream()
{body}
g.replay(); g.replay(); torch.cuda.synchronize()
print('ok', float(a[0]), float(b[0]), float(c[0]))
'''
if __name__ == '__main__':
  for name, body in CASES.items():
    src = TEMPLATE.format(body='\n'.join('  ' + l for l in body.split('\n')))
    r = subprocess.run([sys.executable, '-c', src], capture_output=True, text=True)
    print(name, 'rc', r.returncode, (r.stdout.strip() or r.stderr.strip().split('\n')[-1])[:200])
