#!/bin/bash
# usage (on the GPU box, from the repo root): scripts/profile_round.sh <tag> [bench flags]
# rocprofv3 kernel statistics and PMC passes of the DEFAULT bench command (rocket N=20, 4096 seeds from the script's x0, closed-loop steps 0..29, 3 slices)
# and of its --slices 1 variant, written under gpurun_out/<tag>/.  Counters are collected in their own passes (never combined with a trace domain);
# the program itself follows `--`.  The timed region of a pass = the LAST launches of the dominant kernel (k_rti_chain), as many as the rounds the pass's
# own bench line reports (config.rounds_per_slice): the passes run with --no-cpu --no-secondary, so nothing follows the timed region.
tag=${1:-prof}
shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
steps=${STEPS:-30}
warm=${WARMUP:-1}
cmd="python3 $root/bench.py --steps $steps --warmup $warm --no-cpu --no-secondary $@"
for sl in 3 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s$sl -o s$sl -- $cmd --slices $sl > $out/stats_s$sl.log 2>&1 || echo "stats pass (slices $sl) failed"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/pmc_s$sl/$c -o p -- $cmd --slices $sl > $out/pmc_s${sl}_$c.log 2>&1 || echo "pmc $c (slices $sl) failed"
  done
  echo "slices $sl passes done" >> $out/progress.log
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/pmc_sq -o p -- $cmd --slices 1 > $out/pmc_sq.log 2>&1 || echo "sq pass failed"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/pmc_lds -o p -- $cmd --slices 1 > $out/pmc_lds.log 2>&1 || echo "lds pass failed"
echo "sq passes done" >> $out/progress.log
cd $root
build=$(git rev-parse --short HEAD 2>/dev/null || echo worktree)
python3 scripts/pmc_sq.py $out/pmc_lds $out/pmc_lds.json "$cmd --slices 1" > $out/pmc_lds_summary.txt 2>&1
for sl in 3 1; do
  f=$(ls $out/stats_s$sl/*kernel_stats.csv $out/stats_s$sl/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $out/kernel_stats_s$sl.csv
  # per-dispatch durations of the dominant kernel
  t=$(ls $out/stats_s$sl/*kernel_trace.csv $out/stats_s$sl/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$t" ] && python3 - "$t" $out/k_rti_chain_dispatches_s$sl.csv $out/stats_s$sl.log $out/kernel_time_s$sl.json $sl <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_rti_chain" in r["Kernel_Name"] or "k_qp_solve" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    f.write("index,kernel,start_ns,duration_ns,grid_x\n")
    t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
    for i, r in enumerate(rows):
        nm = "k_rti_chain" if "k_rti_chain" in r["Kernel_Name"] else "k_qp_solve"
        f.write(f"{i},{nm},{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])},{r.get('Grid_Size_X', r.get('Grid_Size', ''))}\n")
# the timed region of this pass: the last sum(rounds) launches of k_rti_chain, rounds from the bench line the pass printed
try:
    line = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    rounds = line["config"].get("rounds_per_slice") or [line["steps"]] * int(sys.argv[5])
    chain = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "k_rti_chain" in r["Kernel_Name"]]
    n = sum(rounds)
    json.dump({"rounds_per_slice": rounds, "timed_region_launches": n, "k_rti_chain_mean_ms_timed_region": sum(chain[-n:]) / max(1, len(chain[-n:])) / 1e6,
               "bench_ms_per_step_under_profiler": line["ms_per_step"], "bench_roofline_avg_launch_ms": line["roofline"]["avg_launch_ms"]}, open(sys.argv[4], "w"), indent=1)
except Exception as e:
    print("kernel_time summary failed:", e)
PY
  python3 scripts/pmc_traffic.py $out/pmc_s$sl $out/pmc_traffic_s$sl.json "$build" "$cmd --slices $sl" 0 $sl "$@" > /dev/null
done
python3 scripts/pmc_sq.py $out/pmc_sq $out/pmc_sq.json "$cmd --slices 1" > $out/pmc_sq_summary.txt 2>&1
# keep what is copied back small: the raw per-dispatch csv files stay on the box
rm -rf $out/stats_s3 $out/stats_s1 $out/pmc_s3 $out/pmc_s1 $out/pmc_sq $out/pmc_lds
ls -la $out
head -12 $out/kernel_stats_s3.csv
