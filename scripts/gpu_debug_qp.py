import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np
from problems import *
np.set_printoptions(linewidth=200, precision=3)
for model, amps in [("pendulum",(0.2,1.0)),("quadrotor",(0.2,1.0,2.0)),("rocket",(0.2,1.0,2.0))]:
    insts=[make_instance(model,s,a) for a in amps for s in range(3)]
    f=make_gpu_solver(insts); push_instances(f,insts)
    lu=[qp1_bounds(i) for i in insts]
    f.qp_update_data_vec(stack(insts,"q"),np.stack([x[0] for x in lu]),np.stack([x[1] for x in lu]))
    x,y,st,it,t=f.qp_solve()
    kkt=f.get("kkt",(8,))
    print(model,'time %.3f ms'%(t*1e3))
    for b in range(len(insts)):
        print('  inst',b,'status',st[b],'its',it[b],'kkt',kkt[b])
    f.close()
