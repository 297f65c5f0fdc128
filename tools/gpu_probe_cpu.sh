cd $GRAFT_REPO_ROOT
python - <<'P'
import os, time, json, sys
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
try:
    print(open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print('no cgroup cpu.max', e)
print(os.popen('lscpu | head -20').read())
sys.argv=['bench.py']
import bench
class A: size=256; model='tapir'; frames=48; queries=256; cpu_sample_frames=12; cpu_sample_queries=48
kw=dict(pyramid_level=0, extra_convs=False, softmax_temperature=20.0)
t=time.time(); r=bench.reference_torch_cpu(A,kw,None,None,None, budget_s=100.0); print(json.dumps(r)[:700]); print('took',time.time()-t, 'cores', bench.host_cores())
P
