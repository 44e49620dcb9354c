"""QP-only micro-benchmark (one k_qp launch over B rocket QPs) for profiling runs."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from robust_nonlinear_mpc_amd import BatchedFastSLS, make_batch
from robust_nonlinear_mpc_amd.fast_sls import DeviceBatch
import ctypes as C
from robust_nonlinear_mpc_amd import _lib as L
B=int(sys.argv[1]) if len(sys.argv)>1 else 4096
reps=int(sys.argv[2]) if len(sys.argv)>2 else 2
eps=float(os.environ.get("QP_EPS","1e-8")); warm=int(os.environ.get("QP_WARM","1"))
model=sys.argv[3] if len(sys.argv)>3 else "rocket"
fx={"rocket":"sweep_rocket_N20_s0.npz","quadrotor":"sweep_quadrotor_N20_s0.npz","pendulum":"sweep_pendulum_N10_s0.npz"}[model]
batch=make_batch(model, os.path.join(ROOT,"tests","golden",fx), B, seed=1)
m,N=batch["model"],batch["N"]
f=BatchedFastSLS(N,m.Q,m.R,m,m.Qf,m.Q_reg,m.R_reg,m.Q_reg_f,batch=B)
f.set_rti_steps(1); f.opts.qp_eps=eps; f.opts.warm_start=warm
dev=DeviceBatch(f,batch)
dev.step()   # sets up tightened bounds too
for r in range(reps):
    x,y,st,it,t=f.qp_solve()
    print("qp launch %.3f ms  its mean %.2f max %d  solved %.4f polished %.4f"%(t*1e3, it.mean(), it.max(), np.mean((st==0)|(st==4)), np.mean(st==0)))
