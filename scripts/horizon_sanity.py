import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from problems import make_instance, run_gpu_fastsls
for model,N,B in (("rocket",32,5),("quadrotor",32,3),("rocket",3,7)):
    insts=[make_instance(model,s,0.5,N=N) for s in range(B)]
    out=run_gpu_fastsls(insts,rti_steps=1)
    print(model,N,B,"success",out["success"].tolist(),"status",out["status"].tolist(),"kkt max",out["kkt"][:,:3].max(),"backoff max %.3e"%out["backoff"].max())
