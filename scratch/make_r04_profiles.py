#!/usr/bin/env python3
"""profiles/r04_* from the files scratch/profile_r04.sh + scratch/timeline_r04.sh + bench.py left under gpurun_out/."""
import csv, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, "gpurun_out"), os.path.join(R, "profiles")
shutil.copy(os.path.join(G, "r04_kernel_stats.csv"), os.path.join(P, "r04_kernel_stats.csv"))
shutil.copy(os.path.join(G, "r04_pmc.json"), os.path.join(P, "r04_pmc_all_kernels.json"))
shutil.copy(os.path.join(G, "r4_timeline.txt"), os.path.join(P, "r04_timeline.txt"))
line = open(os.path.join(G, "r04_bench_final.json")).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(P, "r04_bench.json"), "w").write(line + "\n")
pm = json.load(open(os.path.join(G, "r04_pmc.json")))


def find(prefix):
    k = [k for k in pm if prefix in k]
    assert k, prefix
    return pm[k[0]]


sel = [("rpe_score_kernel", "rpe_score_kernel<2, true>", "one RPE layer over 64 clouds: indices + positions read, geometric score term written", 12608 * 197 * (16 + 4) + 12608 * (1024 + 128) * 4 + 12608 * 4 * 200 * 4),
       ("token_block_kernel<1> (sam6d_linattn_layer)", "token_block_kernel<1,", "one dense LinearTransformerLayer over 64 clouds x 2048 tokens: D read once + D' written once + kv / weight images", 273842176),
       ("token_block_kernel<0> (sam6d_token_block, 12608 / 6304 rows)", "token_block_kernel<0,", "layer tail", None),
       ("xattn_kernel<true> (sam6d_cross_attention_kv)", "xattn_kernel<true>", "cross attention incl. k / v projection", None),
       ("sattn_kernel (sam6d_rpe_self_attention)", "sattn_kernel", "q.k^T + G, softmax, P.v per (cloud, head): q | k | v rows and the score term read, hidden written", 12608 * (768 + 800 + 256) * 4),
       ("rpe_front_kernel", "rpe_front_kernel", "qkv + folds", None), ("rpe_listed_kernel", "rpe_listed_kernel", "listed pairs", None),
       ("fm_sim_kernel", "fm_sim_kernel", "fine-match pipeline", 671613952), ("fm_labels_kernel", "fm_labels_kernel", "fine-match pipeline", None),
       ("fm_assign_kernel", "fm_assign_kernel", "fine-match pipeline", None),
       ("out_split_kernel (sam6d_linear_norm_split)", "out_split_kernel", "out_proj + normalize + operand split: D read once, fp16 hi / lo written", 2 * 32 * 2049 * 256 * 8),
       ("fm_bg_kernel", "fm_bg_kernel", "fine-match pipeline", None), ("fm_merge_sums_kernel", "fm_merge_sums_kernel", "fine-match pipeline", None),
       ("fm_merge_labels_kernel", "fm_merge_labels_kernel", "fine-match pipeline", None),
       ("score_hyp_mfma_kernel", "score_hyp_mfma_kernel", "hypothesis scoring", None), ("pe_mlp_max_h3_kernel", "pe_mlp_max_h3_kernel", "PE MLP + max", None),
       ("tb_kv_fused_kernel", "tb_kv_fused_kernel", "kv side of the dense layer", None)]
out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes; bench.py --steps 2 --warmup 1 --cpu-proposals 0 "
                 "--no-ism --no-config5) on 1x MI355X, final tree of round 4 (scratch/profile_r04.sh); FETCH_SIZE doubled as MI355X_MICROARCH.md "
                 "prescribes for gfx950 (it reports half of a wide coalesced read; Infinity-Cache hits are counted); KB -> bytes x1024; all "
                 "kernels: r04_pmc_all_kernels.json", "kernels": {}}
fm = 0
for name, pref, note, alg in sel:
    e = find(pref)
    d = {"FETCH_SIZE_KB": round(e["FETCH_SIZE"], 1), "WRITE_SIZE_KB": round(e["WRITE_SIZE"], 1), "hbm_bytes_per_launch": e["hbm_bytes_per_launch"],
         "launches_sampled": e["launches_sampled"], "note": note, "mfma_busy_frac": round(e.get("mfma_busy_frac", 0.0), 3)}
    if alg:
        d["algorithmic_bytes_per_launch"] = alg
        d["ratio"] = round(e["hbm_bytes_per_launch"] / alg, 3)
    out["kernels"][name] = d
    if name.startswith("fm_"):
        fm += e["hbm_bytes_per_launch"]
out["fine_match_total_bytes"] = fm
json.dump(out, open(os.path.join(P, "r04_traffic.json"), "w"), indent=1)

rows = list(csv.DictReader(open(os.path.join(P, "r04_kernel_stats.csv"))))
steps = 23.0  # 3 warm-up + 10 timed + 2 x 5 instrumented eager steps
bench = json.loads(line)


def short(n):
    n = n.replace("void ", "")
    for a, b in (("_Z17gemm_nt_h3_kernelILi128ELi128ELb1E", "gemm_nt_h3_kernel<128,128,pre-split W>"), ("_Z17gemm_nt_h3_kernelILi64ELi64ELb0E", "gemm_nt_h3_kernel<64,64>"),
                 ("_Z17gemm_nt_h3_kernelILi64ELi64ELb1E", "gemm_nt_h3_kernel<64,64,pre-split W>"), ("_Z13fm_sim_kernel", "fm_sim_kernel"),
                 ("_Z14fm_prep_kernel", "fm_prep_kernel"), ("_Z12fm_bg_kernel", "fm_bg_kernel")):
        if n.startswith(a):
            return b
    return n.split("(")[0]


o = ["# Per-kernel table at the end of round 4 (1 x MI355X, B = 32, config 2)", "",
     "`scratch/profile_r04.sh`: `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --cpu-proposals 0 --no-ism --no-config5` (23 steps",
     "incl. warm-up and the 2 x 5 instrumented steps: `r04_kernel_stats.csv`), and separate `--pmc` passes (FETCH_SIZE | WRITE_SIZE | MFMA busy | SQ wait",
     "counters; `bench.py --steps 2 --warmup 1`: `r04_pmc_all_kernels.json`).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8);",
     "HBM MB = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch (MI355X_MICROARCH.md); wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES (share of the wave",
     "cycles parked at s_waitcnt / barriers).  Durations are under the profiler.", "",
     "| kernel | launches / step | us / launch | ms / step | MFMA busy | HBM MB / launch | TB/s | wait share |", "|---|---|---|---|---|---|---|---|"]
for r in rows[:40]:
    name = r["Name"]
    calls = int(r["Calls"]) / steps
    avg = float(r["AverageNs"]) / 1e3
    if name.startswith("void at::") or name.startswith("__amd") or "tb_pack" in name or "split_f16" in name:
        continue
    key = [k for k in pm if name[:88] == k[:88] or k.startswith(name[:60])]
    e = pm[key[0]] if key else {}
    busy = ("%.1f %%" % (100 * e["mfma_busy_frac"])) if e.get("mfma_busy_frac", 0) > 0.005 else "-"
    mb = e.get("hbm_bytes_per_launch")
    o.append("| `%s` | %.1f | %.1f | %.3f | %s | %s | %s | %s |" % (short(name), calls, avg, calls * avg / 1e3, busy, ("%.0f" % (mb / 1e6)) if mb else "-",
                                                                   ("%.2f" % (mb / 1e12 / (avg * 1e-6))) if mb else "-",
                                                                   ("%.0f %%" % (100 * e["SQ_WAIT_ANY"] / e["SQ_WAVE_CYCLES"])) if "SQ_WAVE_CYCLES" in e else "-"))
ms = bench["ms_per_step"]
o += ["", "Whole step: %.2f ms = %.0f proposals/s (`r04_bench.json`, the box of the final run); 70.3 GFLOP (fp32-equivalent, folded" % (ms, bench["value"]),
      "formulation, SURVEY 8d) x 32 / %.2f ms = %.0f TFLOP/s of algorithmic work = %.2f of the split-precision bound (2500 / 3 TFLOP/s)." % (ms, 70.3 * 32 / ms, 70.3 * 32 / ms / 833.3),
      "`r04_timeline.txt`: start / duration / queue (q0 main stream, q1 side stream) of every kernel of one step."]
open(os.path.join(P, "r04_stage_table.md"), "w").write("\n".join(o) + "\n")
print("\n".join(o[9:45]))
