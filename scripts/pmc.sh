#!/bin/bash
# usage: scripts/pmc.sh <outdir> <cmd...>   -- separate rocprofv3 --pmc passes (never combined with trace domains)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/$out/p$i -- "$@" > gpurun_out/$out/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import glob,csv,collections
for d in sorted(glob.glob("gpurun_out/$out/p*/")):
    for f in glob.glob(d+"**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        for k,v in acc.items():
            if "k_qp" in k or "k_sweep" in k: print(d.split("/")[-2], k, dict(v))
PY
