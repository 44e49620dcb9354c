set -o pipefail
o=gpurun_out/r03q; mkdir -p $o
run() { name=$1; shift; timeout -k 10 400 "$@" > $o/$name.json 2> $o/$name.err || echo "$name failed"; echo "$name done" >> $o/progress.log; }
run bench_line python bench.py
run bench_line_rounds python bench.py --decoupled 1 --slices 3 --no-cpu
run bench_line_stepwise python bench.py --decoupled 0 --no-cpu --no-secondary
SLSQP_FUSE_RTI=0 run bench_line_separate_launches python bench.py --decoupled 0 --no-cpu --no-secondary
run bench_line_config5_1024 python bench.py --config 5 --no-secondary
run bench_line_pendulum_b1024 python bench.py --model pendulum --batch 1024 --steps 60 --no-secondary
run bench_line_quadrotor_b2048 python bench.py --model quadrotor --batch 2048 --no-secondary
run bench_line_mixed python bench.py --precision 1 --no-secondary --no-cpu
run bench_line_2ranks_gloo python bench.py --gpus 2 --backend gloo --batch 2048 --no-cpu --no-secondary
run bench_line_steps20_warmup5 python bench.py --gpus 1 --steps 20 --warmup 5
timeout -k 10 300 python scripts/latency_b1.py > $o/latency_b1.txt 2>/dev/null || echo latency failed
for f in $o/bench_line*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d["ms_per_step"],2), round(d["value"]), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
