"""B = 1 rocket RTI step in a loop (for `rocprofv3 --kernel-trace`): where does the single-instance latency go -- kernels or gaps?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from problems import make_instance, make_gpu_solver, push_instances, stack
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
insts = [make_instance("rocket", s, 0.5) for s in range(B)]
f = make_gpu_solver(insts)
f.set_rti_steps(1)
f.opts.warm_start = 0
x0 = stack(insts, "x0_arg")
push_instances(f, insts)
for rep in range(8):
    t0 = time.perf_counter()
    f.solve(x0 * (1.0 if rep % 2 == 0 else -1.0), fetch=False)
    print(f"rep {rep}: solve wall {1e3 * (time.perf_counter() - t0):.3f} ms, GPU {f.timing_ms()}", flush=True)
f.close()
