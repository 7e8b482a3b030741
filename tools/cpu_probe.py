import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import nets as O
print('cpu_count', os.cpu_count(), 'affinity', len(os.sched_getaffinity(0)))
try:
    print(open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print('no cpu.max', e)
oG, oD, ooptG, ooptD = O.make_pair('cifar10', 'ns', seed=1)
x = torch.rand(64, 3, 32, 32) * 2 - 1
for nt in (16, 32, 64, 128):
    torch.set_num_threads(nt)
    t0 = time.time(); oD.train_step((x, None), oG, ooptD); t1 = time.time()
    oD.train_step((x, None), oG, ooptD); t2 = time.time()
    print(nt, 'threads: D step', round(t1 - t0, 2), round(t2 - t1, 2), flush=True)
