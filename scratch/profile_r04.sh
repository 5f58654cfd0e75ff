#!/bin/bash
# Round-3 profile set: kernel stats of the default bench command + PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy), each in its own
# rocprofv3 run (kernel-trace only beside --pmc).  Raw output stays under /tmp on the GPU box; the summaries go to gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --steps 10 --warmup 3 --cpu-proposals 0 --no-ism --no-config5 --no-extras --sustained-seconds 0"
rm -rf /tmp/p4 && mkdir -p /tmp/p4
rocprofv3 --kernel-trace --stats -d /tmp/p4/stats -o out --output-format csv -- $CMD > /tmp/p4/stats.log 2>&1
cp /tmp/p4/stats/out_kernel_stats.csv $R/gpurun_out/r04_kernel_stats.csv 2>/dev/null || find /tmp/p4/stats -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r04_kernel_stats.csv \;
tail -1 /tmp/p4/stats.log | cut -c1-400 > $R/gpurun_out/r04_bench_under_rocprof.json
CMD2="python3 $R/bench.py --steps 2 --warmup 1 --cpu-proposals 0 --no-ism --no-config5 --no-extras --sustained-seconds 0"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $set -d /tmp/p4/$tag -o out --output-format csv -- $CMD2 > /tmp/p4/$tag.log 2>&1
done
python3 - <<'PY'
import csv, glob, json, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/p4/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    e = {c: sum(v) / len(v) for c, v in d.items()}
    e["launches_sampled"] = max(len(v) for v in d.values())
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # MI355X_MICROARCH.md: both counters in KB; gfx950 reports half of a wide coalesced read -> FETCH_SIZE doubled
        e["hbm_bytes_per_launch"] = int((2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE", 0) > 0:
        e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (e["GRBM_GUI_ACTIVE"] / 8.0)
    out[k[:90]] = e
json.dump(out, open(R + "/gpurun_out/r04_pmc.json", "w"), indent=1, sort_keys=True)
print("kernels with counters:", len(out))
PY
python3 $R/scratch/kstats.py $R/gpurun_out/r04_kernel_stats.csv 10 45
