"""Latency of the single-instance drop-in (fast_SLS, B = 1: what the reference's closed-loop scripts call once per SCP iteration) next to the CPU
restatement on one host thread.  One RTI fast-SLS step = update_dynamics_list + update_linear_cost + solve (2 QPs + 1 sweep), host buffers in,
result dict out.  Also B = 8 / 64 to show where the batch starts to pay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from problems import make_instance, run_oracle_fastsls, make_gpu_solver, push_instances, stack
from oracle import oracle as O
for model in ("pendulum", "quadrotor", "rocket"):
    for B in (1, 8, 64):
        insts = [make_instance(model, s, 0.5) for s in range(B)]
        f = make_gpu_solver(insts)
        f.set_rti_steps(1)
        f.opts.warm_start = 0
        x0 = stack(insts, "x0_arg")
        ts, tg = [], []
        for rep in range(12):
            t0 = time.perf_counter()
            push_instances(f, insts)
            f.solve(x0 * (1.0 if rep % 2 == 0 else -1.0), fetch=False)
            pv = f.get("primal_vec", (f.n,)); bx = f.get("backoff_x", (insts[0].N + 1, insts[0].m.nx))
            ts.append(time.perf_counter() - t0); tg.append(f.timing_ms()["total"])
        st = f.get("status", (), np.int32)
        f.close()
        line = f"{model:9s} B={B:3d}: GPU wall {1e3*np.median(ts[2:]):7.2f} ms per RTI step of the batch (GPU time {np.median(tg[2:]):6.2f} ms), certified {np.mean(st == 0):.2f}"
        if B == 1:
            t0 = time.perf_counter()
            for rep in range(3):
                run_oracle_fastsls(insts[0], rti_steps=1, settings=O.default_settings())
            line += f" | CPU restatement (OSQP-class, default settings + polish), 1 thread: {1e3*(time.perf_counter()-t0)/3:7.2f} ms"
        print(line, flush=True)
