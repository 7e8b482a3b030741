import os, sys
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
def maps(tag):
    libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l or 'hsa-runtime' in l or 'libdiagan' in l})
    print(tag, libs, flush=True)
print("torch loaded before build:", 'torch' in sys.modules)
g.build()
print("torch loaded after build:", 'torch' in sys.modules)
maps("after build")
import torch
print(torch.cuda.is_available(), torch.cuda.device_count())
torch.cuda.set_device(0)
x = torch.ones(4, device='cuda')
print(x.sum().item())
maps("after torch init")
from diagan import _native as nat
import ctypes
try:
    from diagan.ops import eltwise as E
    y = E.tanh_fwd(torch.zeros(1, 2, 2, 4, device='cuda'))
    print("native call ok", y.sum().item())
except Exception as e:
    print("native call failed:", e)
