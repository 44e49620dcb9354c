"""Which instances make the tail of the persistent closed-loop launch?  Per instance: block solves summed over the run, how often its QPs fell back to the
interior point, how persistent that is from one MPC step to the next.  usage: tail_probe.py [B] [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robust_nonlinear_mpc_amd import ClosedLoopMPC, disturbance_stream, get_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
m = get_model("rocket"); N = 20
x0 = np.tile(m.extra["x0"], (B, 1))
W = np.stack([disturbance_stream(s, steps, m.nx) for s in range(B)], axis=1)
cl = ClosedLoopMPC(m, N, B)
cl.f.opts.time_kernels = 1
out = cl.run_decoupled(x0, steps, W, solve_nominal=True, continuation=2)
qs = out["qp_stats"]                       # (B, steps, 2, 8): its, blk, fac, nact, warm, rounds, status, path
blk = qs[..., 1].astype(float)
tot = blk.sum(axis=(1, 2))
print("launch ms", out["loop_stats"]["launch_ms"], "block solves per instance over the run: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max()))
order = np.argsort(-tot)
fb = (qs[..., 7] % 10 > 0)                  # the solve left its first attempt (interior point / second attempt)
ran = (qs[..., 6] != -1) & (qs[..., 6] != 2)
print("fallback rate overall: qp1 %.4f qp2 %.4f" % (fb[..., 0][ran[..., 0]].mean(), fb[..., 1][ran[..., 1]].mean()))
for slot in (0, 1):
    f, r = fb[:, :, slot], ran[:, :, slot]
    prev, cur = f[:, :-1], f[:, 1:]
    ok = r[:, :-1] & r[:, 1:]
    p11 = cur[ok & prev].mean() if (ok & prev).any() else float("nan")
    p01 = cur[ok & ~prev].mean() if (ok & ~prev).any() else float("nan")
    print(f"qp{slot + 1}: P(fallback | fallback in the previous step) {p11:.3f}, P(fallback | none) {p01:.3f}; block solves with fallback mean {blk[:, :, slot][f & r].mean():.1f}, without {blk[:, :, slot][~f & r].mean():.1f}")
print("the 12 heaviest instances: total block solves, steps run, fallbacks qp1/qp2, block solves per step")
for b in order[:12]:
    print(f"  seed {b:5d}: {tot[b]:6.0f}  steps run {int(ran[b, :, 0].sum()):2d}  fallbacks {int((fb[b, :, 0] & ran[b, :, 0]).sum()):2d}/{int((fb[b, :, 1] & ran[b, :, 1]).sum()):2d}  per step " + " ".join(f"{int(v)}" for v in blk[b].sum(axis=1)))
its = qs[..., 0]
print("interior-point iterations when used: qp1 mean %.1f, qp2 mean %.1f" % (its[..., 0][its[..., 0] > 0].mean(), its[..., 1][its[..., 1] > 0].mean()))
# how much of a fallback solve is the failed first attempt?  rounds (index 5) counts active-set rounds of the attempt(s)
rd = qs[..., 5].astype(float)
print("fallback solves: active-set rounds mean qp1 %.1f qp2 %.1f" % (rd[..., 0][fb[..., 0] & ran[..., 0]].mean(), rd[..., 1][fb[..., 1] & ran[..., 1]].mean()))
heavy = (blk > 40) & ran
for slot in (0, 1):
    hv, r, f = heavy[:, :, slot], ran[:, :, slot], fb[:, :, slot]
    ok = r[:, :-1] & r[:, 1:]
    for nm, prev in (("heavy", hv[:, :-1]), ("fallback", f[:, :-1]), ("two fallbacks in a row", np.concatenate([np.zeros((B, 1), bool), f[:, :-2] & f[:, 1:-1]], axis=1))):
        sel = ok & prev
        if sel.any():
            print(f"qp{slot + 1}: after a previous-step {nm} solve ({int(sel.sum())} cases): P(fallback) {f[:, 1:][sel].mean():.3f}, P(heavy) {hv[:, 1:][sel].mean():.3f}, block solves mean {blk[:, 1:, slot][sel].mean():.1f}; "
                  f"of all heavy solves {int((hv[:, 1:] & sel).sum())} / {int(hv.sum())} follow one")
    # same-step coupling: QP2 after a QP1 that fell back
sel = ran[:, :, 0] & ran[:, :, 1] & fb[:, :, 0]
print(f"qp2 in a step whose qp1 fell back ({int(sel.sum())} cases): P(fallback) {fb[:, :, 1][sel].mean():.3f}, block solves mean {blk[:, :, 1][sel].mean():.1f}")
for slot in (0, 1):
    hq = qs[:, :, slot, :][heavy[:, :, slot]]
    if len(hq):
        print(f"qp{slot + 1} solves with more than 40 block solves: {len(hq)}; block solves mean {hq[:, 1].mean():.1f}, interior-point iterations mean {hq[:, 0].mean():.1f} max {hq[:, 0].max()}, "
              f"factorisations mean {hq[:, 2].mean():.1f}, active-set rounds mean {hq[:, 5].mean():.1f} max {hq[:, 5].max()}, status counts {np.bincount(np.clip(hq[:, 6], 0, 7), minlength=6).tolist()}, path counts {np.bincount(hq[:, 7] % 10, minlength=4).tolist()}")
        for r in hq[np.argsort(-hq[:, 1])[:8]]:
            print("     its %d blk %d fac %d nact %d warm %d rounds %d status %d path %d" % tuple(r))
cl.close()
