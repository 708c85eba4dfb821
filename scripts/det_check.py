import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tests.test_gpu_deterministic import _run
cuda=torch.device("cuda")
for dt in (torch.float32, torch.bfloat16):
    a=_run(cuda,dt,False); b=_run(cuda,dt,False)
    nd=sum(int((a[1][k]!=b[1][k]).sum()) for k in a[1]); tot=sum(v.size for v in a[1].values())
    print(dt, "default mode, two runs: differing elements", nd, "of", tot, "losses equal", a[0]==b[0])
