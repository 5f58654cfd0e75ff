#!/bin/bash
# SQ wait / active counters of the latency-bound sparse-stage kernels (micro-benchmark loop of scratch/ub.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_tb_$tag -o out --output-format csv -- python3 $R/scratch/ub.py token_block_12608 cross_layer > /tmp/pmc_tb_$tag.log 2>&1
done
find /tmp/pmc_tb_* -name '*.csv' | head -20; tail -3 /tmp/pmc_tb_SQ_WAVE_CYCLES.log
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("/tmp/pmc_tb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "token_block" in k or "xattn" in k or "gemm" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
