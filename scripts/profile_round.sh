#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag> [bench flags]
# rocprofv3 kernel statistics and PMC passes of the DEFAULT bench command (rocket N=20, 4096 seeds from the script's x0, closed-loop steps 0..29 as ONE
# persistent launch of k_cl_loop: variant "p") and of the round-based loop on three slices (--decoupled 1 --slices 3, k_rti_chain: variant "r"; skipped with
# VARIANTS=p), written under gpurun_out/<tag>/.  Counters are collected in their own passes (never combined with a trace domain); the program itself follows
# `--`.  The timed region of a pass = the LAST launches of the dominant kernel, as many as the pass's own bench line reports (config.rounds_per_slice; the
# persistent launch: one): the passes run with --no-cpu --no-secondary, so nothing follows the timed region.
tag=${1:-prof}
shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
steps=${STEPS:-30}
warm=${WARMUP:-1}
variants=${VARIANTS:-p r}
cmd="python3 $root/bench.py --steps $steps --warmup $warm --no-cpu --no-secondary $@"
flags_of() { if [ "$1" = "p" ]; then echo ""; else echo "--decoupled 1 --slices 3"; fi; }
for v in $variants; do
  fl=$(flags_of $v)
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$v -o s$v -- $cmd $fl > $out/stats_$v.log 2>&1 || echo "stats pass ($v) failed"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_$v/$c -o p -- $cmd $fl > $out/pmc_${v}_$c.log 2>&1 || echo "pmc $c ($v) failed"
  done
  echo "variant $v passes done" >> $out/progress.log
done
if [ "${SEPARATE:-0}" = "1" ]; then
  # the separate launches (k_qp_solve / k_after_qp / k_sweep_ric1 / k_sweep_prop / k_tighten / k_qp_solve per step and slice): per-kernel times of the sweep kernels
  export SLSQP_FUSE_RTI=0
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_sep -o ssep -- $cmd --decoupled 0 > $out/stats_sep.log 2>&1 || echo "stats pass (separate launches) failed"
  unset SLSQP_FUSE_RTI
  f=$(ls $out/stats_sep/*kernel_stats.csv $out/stats_sep/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $out/kernel_stats_separate_launches.csv
  rm -rf $out/stats_sep
fi
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_sq -o p -- $cmd > $out/pmc_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/pmc_lds -o p -- $cmd > $out/pmc_lds.log 2>&1 || echo "lds pass failed"
echo "sq passes done" >> $out/progress.log
cd $root
build=$(git rev-parse --short HEAD 2>/dev/null || echo worktree)
python3 scripts/pmc_sq.py $out/pmc_lds $out/pmc_lds.json "$cmd" > $out/pmc_lds_summary.txt 2>&1
for v in $variants; do
  fl=$(flags_of $v)
  f=$(ls $out/stats_$v/*kernel_stats.csv $out/stats_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $out/kernel_stats_$v.csv
  # per-dispatch durations of the dominant kernel
  t=$(ls $out/stats_$v/*kernel_trace.csv $out/stats_$v/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$t" ] && python3 - "$t" $out/dominant_kernel_dispatches_$v.csv $out/stats_$v.log $out/kernel_time_$v.json <<'PY'
import csv, json, sys
NAMES = ("k_cl_loop", "k_rti_chain", "k_qp_solve")
def short(n):
    for k in NAMES:
        if k in n:
            return k
    return None
rows = [r for r in csv.DictReader(open(sys.argv[1])) if short(r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    f.write("index,kernel,start_ns,duration_ns,grid_x\n")
    t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
    for i, r in enumerate(rows):
        f.write(f"{i},{short(r['Kernel_Name'])},{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])},{r.get('Grid_Size_X', r.get('Grid_Size', ''))}\n")
# the timed region of this pass: the last sum(rounds) launches of the dominant kernel, rounds from the bench line the pass printed
try:
    line = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    dom = line["roofline"]["kernel"]
    rounds = line["config"].get("rounds_per_slice") or [line["steps"]] * int(line["config"]["slices_per_gpu"])
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if short(r["Kernel_Name"]) == dom]
    n = sum(rounds)
    json.dump({"kernel": dom, "rounds_per_slice": rounds, "timed_region_launches": n, "mean_ms_timed_region": sum(dur[-n:]) / max(1, len(dur[-n:])) / 1e6,
               "bench_ms_per_step_under_profiler": line["ms_per_step"], "bench_roofline_avg_launch_ms": line["roofline"]["avg_launch_ms"]}, open(sys.argv[4], "w"), indent=1)
except Exception as e:
    print("kernel_time summary failed:", e)
PY
  python3 scripts/pmc_traffic.py $out/pmc_$v $out/pmc_traffic_$v.json "$build" "$cmd $fl" $v > /dev/null
done
python3 scripts/pmc_sq.py $out/pmc_sq $out/pmc_sq.json "$cmd" > $out/pmc_sq_summary.txt 2>&1
# keep what is copied back small: the raw per-dispatch csv files stay on the box
for v in $variants; do rm -rf $out/stats_$v $out/pmc_$v; done
rm -rf $out/pmc_sq $out/pmc_lds
ls -la $out
head -12 $out/kernel_stats_p.csv
