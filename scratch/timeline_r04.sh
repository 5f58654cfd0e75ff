#!/bin/bash
# Kernel timeline of one benchmark step: start / duration / queue of every kernel (rocprofv3 --kernel-trace), condensed to text.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl && mkdir -p /tmp/tl
rocprofv3 --kernel-trace -d /tmp/tl -o out --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --cpu-proposals 0 --no-ism --no-config5 --no-extras --sustained-seconds 0 > /tmp/tl/bench.log 2>&1
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last step: split at fps_reg_kernel launches
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void fps_reg_kernel")]
a = starts[-1]
# step begins a few kernels before FPS (copies / in_proj on the side queue): go back to the previous pick/fine_finish end
while a > 0 and "fine_finish" not in rows[a - 1]["Kernel_Name"] and "procrustes" not in rows[a-1]["Kernel_Name"] and a > starts[-1] - 12:
    a -= 1
seg = rows[a:]
t0 = int(seg[0]["Start_Timestamp"])
queues = {}
out = []
last_end = {}
for r in seg:
    q = queues.setdefault(r["Queue_Id"], len(queues))
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = s - last_end.get(q, s)
    last_end[q] = e
    out.append("%9.1f %8.1f q%d gap %6.1f  %s" % (s / 1e3, (e - s) / 1e3, q, gap / 1e3, r["Kernel_Name"][:70]))
open(R + "/gpurun_out/r4_timeline.txt", "w").write("\n".join(out) + "\n")
print("kernels in the step:", len(seg), "span %.1f us" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e3))
PY
tail -2 /tmp/tl/bench.log | cut -c1-200
