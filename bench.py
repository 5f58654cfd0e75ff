#!/usr/bin/env python3
"""bench.py -- proposals/sec through the PEM match+SVD path (BASELINE.json metric) on N MI355X GPUs of one node.

A "step" is one pass of the matching path (FPS x2 -> geo-embedding x2 -> CoarsePointMatching -> FinePointMatching ->
R, t, score; the region PEM/model/pose_estimation_model.py:29-55 runs after feature extraction) over one batch of synthetic
proposals with inputs resident in HBM when the timed region starts.

  --workload config2 (default, the headline): B = 32 proposals per GPU (2048 scene + 2048 template points, 1024
      CAD points, random-init weights).  N > 1: every rank its own 32 proposals (weak scaling).
  --workload config4: ONE LM-O-style scene of 200 proposals of 8 objects, dealt round-robin to the N ranks
      (strong scaling: 200 / N proposals per rank, 25 at N = 8), unique templates + template_ids per rank.

One process per GPU; the ranks all-gather the 13 floats / proposal (R, t, score) over RCCL inside the timed region.  Small batches
(fewer than 16 proposals per GPU) replay the step from a hipGraph (sam6d_hip.pem.PemGraph: one graph launch instead of ~250 kernel
launches); at B = 32 the eager launch sequence is the faster one and is what is timed (--graph 1 / 0 force either).

Prints ONE JSON line (rank 0).  Beside the contract's keys:
  roofline          the dominant kernel (rpe_score_kernel): executed fp16-MFMA flops / dense fp16 peak, launch durations from HIP events
                    around the kernel in eager steps right behind the timed region (events cannot be recorded inside a captured graph);
                    the three-product yardstick of rounds 1-3 as `frac_three_product_yardstick`
  sustained         the same step for >= 3 s of continuous replays after the timed K steps (the chip lowers its clock under load)
  paths             which weight-guarded kernel routes this weight set takes
  per_rank_ms, allgather_ms   (N > 1) every rank's own time per step and the all-gather's
  latency           B = 1 and B = 25 (one proposal: config 1's shape; a strong-scaled shard of config 4), eager and graph
  config4           (config2 runs, N = 1) the 200-proposal scene in one call on this GPU
  fallbacks         the cost of the routes a different weight set may take (three-product stage 1, materialised embedding)
  cpu_baseline      the CPU oracle (oracle/pem_oracle.py) on this box's host cores on a bounded sample, with the CPU model
  ism_config3, config5   the ISM leg and the 4096-point shape, reported beside the headline
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "openvino-sam-6d_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

B_PER_GPU = 32
CONFIG4_TOTAL, CONFIG4_OBJECTS = 200, 8
GEO_FLOP_PER_CLOUD = 2.0 * 197 * 197 * 4 * 256 * 256  # 20.35 GFLOP: (d + 3 angular rows) x 256x256 per pair
PROJP_FLOP_PER_CLOUD = 2.0 * 197 * 197 * 256 * 256  # 5.09 GFLOP: proj_p of the embedding, per RPE layer
# what rpe_score_kernel itself contracts per query token: 3 angular rows x 197 keys x 256 channels x 32 Chebyshev
# orders, the 4 head dots over 256 channels, the 32-term d part for 4 heads -- fp32-equivalent flops, unpadded
RPE_FLOP_PER_QUERY = 2.0 * 197 * (3 * 256 * 32 + 4 * 256 + 4 * 32)
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_FP16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16/bf16 MFMA
PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
TRAFFIC_FILE = os.path.join("profiles", "r04_traffic.json")  # HBM bytes per launch from separate rocprofv3 --pmc passes (recorded, not live)
TRAFFIC_FALLBACK = os.path.join("profiles", "r03_traffic.json")
CPU_CHUNK = 1  # proposals per CPU-baseline call (the reference's sampling step needs ~2.8 GB per proposal)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, nprop, threads):
    """The reference algorithm's CPU port (oracle), including the reference's dense (ns x N) compare in the weighted
    sampling (PEM/utils/model_utils.py:277-305), on `nprop` proposals of the config-2 generator, CPU_CHUNK at a time (the
    reference needs ~2.8 GB per proposal in that step)."""
    from oracle import pem_oracle as O
    from sam6d_hip import synth
    torch.set_num_threads(threads)
    inp = synth.config2_inputs(B=nprop, seed=1)
    outs = []
    t0 = time.perf_counter()
    for i in range(0, nprop, CPU_CHUNK):
        sl = lambda k: inp[k][i:i + CPU_CHUNK].contiguous()
        with torch.no_grad():
            outs.append(O.pem_match(sl("dense_pm"), sl("dense_fm"), sl("dense_po"), sl("dense_fo"), sl("radius"), sl("model"),
                                    sd, sl("rand"), faithful=True))
    dt = time.perf_counter() - t0
    R = torch.cat([o[0] for o in outs]); t = torch.cat([o[1] for o in outs]); s = torch.cat([o[2] for o in outs])
    return nprop / dt, dt, (R, t, s), inp


def launch_ranks(n, argv, script=None, extra_env=None, check_devices=True):
    """Start `n` fresh child processes of `script` (default: this file), one rank per GPU, and wait for them.  Used when
    `--gpus N` (N > 1) is given without a launcher (WORLD_SIZE unset).  The parent makes NO GPU call: it only counts devices
    (torch.cuda.device_count() does not initialise HIP on this image) and hands every child RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT -- the same environment torch.distributed.run provides.  Children inherit stdout (rank 0 prints
    the JSON line).  Returns the first non-zero exit code (the remaining ranks are then terminated by PID), else 0."""
    import socket
    import subprocess
    if check_devices:
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d needs %d visible GPUs, found %d -- refusing to run fewer ranks than asked\n" % (n, n, have))
            return 2
    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for pr in list(pending):
            try:
                code = pr.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            pending.remove(pr)
            if code != 0 and rc == 0:
                rc = code
                for other in pending:  # a failed rank would leave the others waiting in the collective
                    other.terminate()
    return rc


def bench_ism(dev, reps=10):
    """BASELINE config 3 on one GPU (not the headline metric: reported beside it): 200 proposals x 42 templates through the ISM scoring
    path -- class-token cosine + avg-5 selection, patch similarity (batched 256 x 256 x 1024 contraction) with appearance score and
    visible ratio, masked-depth translation + template projection + IoU, final score -- inputs resident in HBM, one host read-back per pass
    (the count of selected proposals, as in the reference's boolean-mask indexing).  Returns the `ism_config3` object."""
    from sam6d_hip import ism, synth
    d = synth.config3_inputs(0)
    g = {k: (v.to(dev).contiguous() if torch.is_tensor(v) else v) for k, v in d.items()}
    masks_u8 = (g["masks"] > 0).to(torch.uint8).contiguous()  # SAM's proposals are binary masks: one byte per pixel, read in place

    def one_pass(timers=None):
        def mark(name):
            if timers is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                timers.append((name, e))
        mark("start")
        sim = ism.pairwise_similarity(g["q"], g["ref"])
        sel, obj, sem, best = ism.semantic_select(sim, "avg_5", 0.2)
        mark("semantic")
        # appearance score + visible ratio in one launch: query and template patch descriptors read in place through their indices
        ps = ism.patch_scores_fused(g["q_appe"], g["r_appe"], obj, best, q_index=sel)
        mark("patch_similarity")
        appe, vis = ps.scores(0.5)
        mark("patch_scores")
        vu, xyxy, tr = ism.project_template_to_image(best, obj, g["poses"], g["pc"], masks_u8, g["depth"], g["K"], g["depth_scale"],
                                                     mask_index=sel)
        iou, flag = ism.compute_iou(xyxy, ism.take_rows(g["boxes"], sel), return_flag=True)
        fin = ism.final_score(sem, appe, iou, vis, all_positive=flag)  # (the IoU quirk decided on the device: no second read-back)
        mark("geometric")
        return fin, len(sel)

    for _ in range(3):
        fin, ns = one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fin, ns = one_pass()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    tm = []
    one_pass(tm)
    torch.cuda.synchronize()
    stages = {tm[i][0]: tm[i - 1][1].elapsed_time(tm[i][1]) for i in range(1, len(tm))}
    Nq, Pn, D = g["q_appe"].shape
    H, Wd = g["masks"].shape[1:]
    flop = 2.0 * ns * Pn * Pn * D
    byts = 2.0 * ns * Pn * D * 4  # both descriptor sets read once; the similarity never leaves the chip
    sim_ms = stages["patch_similarity"]
    geo_bytes = ns * H * Wd * 1.0 + H * Wd * 4.0  # the selected proposals' masks (1 byte / pixel) + the depth map
    return {"workload": "ISM template scoring, %d proposals x %d templates, %d x %d patch descriptors, %d selected by the 0.2 threshold"
                        % (Nq, g["ref"].shape[1], Pn, D, ns),
            "proposals_per_s": Nq / (ms * 1e-3), "ms_per_pass": ms, "stage_ms": stages,
            "roofline_patch_similarity": {"bound": "hbm", "kernel": "ism_patch_fused_kernel (indexed operands, row / column maxima in the epilogue)", "achieved": byts / 1e9 / (sim_ms * 1e-3), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                          "frac": byts / 1e9 / (sim_ms * 1e-3) / PEAK_HBM_GBS, "traffic": None, "launch_ms": sim_ms,
                                          "algorithmic_mb_per_launch": byts / 1e6, "algorithmic_gflop_per_launch": flop / 1e9,
                                          "tflops": flop / 1e12 / (sim_ms * 1e-3)},
            "geometric_stage": {"ms": stages["geometric"], "algorithmic_mb": geo_bytes / 1e6,
                                "gbs": geo_bytes / 1e9 / (stages["geometric"] * 1e-3),
                                "note": "masked-depth translation (uint8 masks read in place through the selection) + projection + IoU + final score"},
            "finite": bool(torch.isfinite(fin).all())}


def bench_config5(dev, W, B=16, steps=5, warmup=2):
    """BASELINE config 5 on ONE GPU's share (reported beside the headline, not part of `value`): 4096 scene + 4096 template points
    (fine_npoint = 4096), B = 16 proposals per GPU (128 over 8 GPUs), every contraction on fp16 MFMA with fp32 accumulation.  The
    arithmetic is the library's DEFAULT fp16 x3 split (three fp16 MFMA products per fp32 product, ~1e-6): the single-product variant
    (matmul mode 2) does not preserve the poses on random-init weights (DESIGN 4 "Mode 2") and is not used.  Rooflines for the two
    kernels that scale with the dense point count: the fused dense linear-attention layer and the fine-match pipeline at n = 4097."""
    from sam6d_hip import pem, synth
    N = 4096
    inp = synth.config2_inputs(B=B, seed=5, n_dense=N)
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}

    def step():
        return pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W, d["rand"])

    for _ in range(warmup):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    pem.PROFILE_NAMES = {"linattn_layer", "fine_match"}
    pem.PROFILE = {}
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    prof, pem.PROFILE, pem.PROFILE_NAMES = pem.PROFILE, None, None
    res = {"workload": "PEM batch=%d proposals/GPU (128 over 8 GPUs), 4096 scene + 4096 model pts, 1024 CAD pts, random-init weights "
                       "(SURVEY 8d config 5)" % B, "proposals_per_gpu": B, "ms_per_step": ms, "proposals_per_s": B / (ms * 1e-3),
           "matmul": "fp16x3 split-precision MFMA, fp32 accumulate (~1e-6 rel.; the 1e-4 pose contract holds)",
           "finite": bool(all(torch.isfinite(o).all() for o in out))}
    lev = [a.elapsed_time(b) for a, b in prof.get("linattn_layer", [])]
    if lev:
        ms1 = sum(lev) / len(lev)
        tok = 2 * B * N
        fl = tok * 2.0 * (2 * 256 * 256 + 256 * 64 + 2 * 256 * 512)
        ach = fl / (ms1 * 1e-3) / 1e12
        res["roofline_dense_layer"] = {"bound": "mfma", "kernel": "token_block_kernel<1>, %d clouds x %d tokens per launch" % (2 * B, N),
                                       "achieved": ach, "peak": PEAK_FP16_MFMA_TFLOPS / 3.0, "unit": "TFLOP/s",
                                       "frac": ach / (PEAK_FP16_MFMA_TFLOPS / 3.0), "traffic": None, "launch_ms": ms1,
                                       "launches_timed": len(lev), "algorithmic_gflop_per_launch": fl / 1e9,
                                       "algorithmic_mb_per_launch": 2 * tok * 256 * 4 / 1e6}
    fev = [a.elapsed_time(b) for a, b in prof.get("fine_match", [])]
    if fev:
        ms1 = sum(fev) / len(fev)
        n = N + 1
        Eb = B * n * (n + 3) * 4
        feat = 2 * B * n * 256 * 4
        alg = feat + feat + feat + Eb + feat + Eb + Eb  # as roofline_fine_match of the headline config
        gbs = alg / 1e9 / (ms1 * 1e-3)
        res["roofline_fine_match"] = {"bound": "hbm", "kernel": "sam6d_fine_match at n = 4097 (chunked label / assignment passes + merges)",
                                      "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                                      "launch_ms": ms1, "launches_timed": len(fev), "algorithmic_mb_per_launch": alg / 1e6}
    return res


def _timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def bench_latency(dev, W, use_graph):
    """Small-batch latency (not part of `value`): B = 1 (config 1's shape: ONE proposal) and B = 25 (a strong-scaled shard of config 4,
    200 / 8).  At these sizes the eager step is bound by the host's launch rate (~250 ctypes launches); the hipGraph replay is one
    launch."""
    from sam6d_hip import pem, synth
    out = {}
    for B in (1, 25):
        inp = synth.config2_inputs(B=B, seed=7)
        d = {k: v.to(dev).contiguous() for k, v in inp.items()}
        args = [d[k] for k in ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")]
        e = {"eager_ms": _timed(lambda: pem.pem_match(*args, W, d["rand"]), 10, 3)}
        if use_graph:
            try:
                g = pem.PemGraph(W, *args, d["rand"], microbatch=1)
                e["graph_ms"] = _timed(g.replay, 20, 3)
                del g
            except Exception as ex:  # a side measurement must not take the headline down
                e["graph_error"] = "%s: %s" % (type(ex).__name__, ex)
        best = min(v for k, v in e.items() if k.endswith("_ms"))
        e["proposals_per_s"] = B / (best * 1e-3)
        out["B%d" % B] = e
    out["note"] = "ms per step (one call over B proposals), inputs resident; eager = one ctypes launch per kernel, graph = one hipGraph replay"
    return out


def bench_config4_single(dev, W, use_graph, mb):
    """Config 4 on THIS GPU (reported beside the config-2 headline): the whole 200-proposal, 8-object scene in one call with
    template_ids (template-side work once per object)."""
    from sam6d_hip import pem, synth
    inp = synth.config4_inputs(B=CONFIG4_TOTAL, n_obj=CONFIG4_OBJECTS, seed=4)
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    args = [d[k] for k in ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")]
    res = {"workload": "one scene: %d proposals of %d objects in one call, unique templates + template_ids (SURVEY 8d config 4)"
                       % (CONFIG4_TOTAL, CONFIG4_OBJECTS)}
    res["eager_ms"] = _timed(lambda: pem.pem_match(*args, W, d["rand"], template_ids=d["template_ids"]), 3, 1)
    if use_graph:
        try:
            g = pem.PemGraph(W, *args, d["rand"], microbatch=mb, template_ids=d["template_ids"])
            res["graph_ms"] = _timed(g.replay, 5, 1)
            del g
        except Exception as ex:
            res["graph_error"] = "%s: %s" % (type(ex).__name__, ex)
    best = min(v for k, v in res.items() if k.endswith("_ms"))
    res["proposals_per_s"] = CONFIG4_TOTAL / (best * 1e-3)
    rep = synth.repeated(d)
    res["repeated_form_eager_ms"] = _timed(lambda: pem.pem_match(*[rep[k] for k in ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius",
                                                                                       "model")], W, rep["rand"]), 3, 1)
    return res


def bench_fallbacks(dev, sd, d, steps=5):
    """What the routes a DIFFERENT weight set may be sent to cost at config 2 (the released checkpoint decides, not this benchmark: DESIGN
    'weight guards'): stage 1 of the RPE scores with three MFMA products, and the materialised embedding instead of the fused score
    kernel.  Eager steps, ms per step; `default_eager_ms` measured the same way beside them."""
    from sam6d_hip import pem
    args = [d[k] for k in ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")]
    out = {}
    for name, over in (("default_eager_ms", {}), ("rpe_three_product_stage1_ms", {"rpe_products": 3}), ("materialised_embedding_ms", {"fused_rpe": False})):
        Wx = pem.PemWeights(sd, dev, options=pem.Options.from_env(**over))
        try:
            out[name] = _timed(lambda: pem.pem_match(*args, Wx, d["rand"]), steps, 2)
        except Exception as ex:
            out[name] = "%s: %s" % (type(ex).__name__, ex)
        del Wx
        torch.cuda.empty_cache()
    return out


def _stub_match(inp):
    """--stub-compute (CPU test hook, tests/test_bench_launcher.py): a deterministic stand-in for pem_match that depends on every
    input of a proposal and on nothing else, so that the sharding / gather / unshard bookkeeping can be checked bit for bit without a GPU."""
    B = inp["dense_pm"].shape[0]
    po = inp["dense_po"][inp["template_ids"]] if "template_ids" in inp else inp["dense_po"]
    fo = inp["dense_fo"][inp["template_ids"]] if "template_ids" in inp else inp["dense_fo"]
    a = inp["dense_pm"].reshape(B, -1).double().sum(1) + 3 * po.reshape(B, -1).double().sum(1)
    b = inp["dense_fm"].reshape(B, -1).double().sum(1) - fo.reshape(B, -1).double().mean(1)
    c = inp["rand"].double().sum(1) + inp["model"].reshape(B, -1).double().sum(1)
    R = torch.stack([a + k * b for k in range(9)], 1).float().reshape(B, 3, 3)
    t = torch.stack([c, a * c, b - c], 1).float()
    return R, t, (a - b + c).float()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=("config2", "config4"), default="config2")
    ap.add_argument("--cpu-proposals", type=int, default=8, help="proposals in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="config2: proposals per GPU")
    ap.add_argument("--graph", type=int, default=-1, help="1: replay the step from a hipGraph; 0: eager launches; -1 (default): the graph "
                    "where the eager step is bound by the host's launch rate (fewer than 16 proposals per GPU), eager otherwise -- measured "
                    "in round 4: at B = 32 the replay is 1-2 %% SLOWER than the eager sequence (7.22 against 7.09 ms), at B = 1 it is what "
                    "removes the launch-rate bound (`latency` in the JSON line)")
    ap.add_argument("--microbatch", type=int, default=0, help="slices of the batch on concurrent streams (0 = the default for the mode)")
    ap.add_argument("--sustained-seconds", type=float, default=3.0, help="continuous steps after the timed region (0 = skip)")
    ap.add_argument("--no-ism", action="store_true", help="skip the ISM (config 3) leg reported beside the PEM metric")
    ap.add_argument("--no-config5", action="store_true", help="skip the 4096-point (config 5) leg reported beside the PEM metric")
    ap.add_argument("--no-extras", action="store_true", help="skip latency / config4 / fallback legs")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for the CPU rehearsal)")
    ap.add_argument("--stub-compute", action="store_true", help="CPU test hook: replace the GPU path by a deterministic stand-in")
    ap.add_argument("--out", default=None, help="also write the JSON line to this file")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: become one -- N fresh children, started before anything in this process touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:], check_devices=not args.stub_compute))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run --nproc-per-node %d (or without a "
                         "launcher: bench.py starts the ranks itself)\n" % (args.gpus, world, args.gpus))
        sys.exit(2)
    stub = args.stub_compute
    if stub:
        dev = torch.device("cpu")
    else:
        if local >= torch.cuda.device_count():
            sys.stderr.write("bench.py: rank %d has no GPU (LOCAL_RANK %d, %d visible)\n" % (rank, local, torch.cuda.device_count()))
            sys.exit(2)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub or args.backend != "nccl":
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == world
    sync = (lambda: None) if stub else torch.cuda.synchronize

    from sam6d_hip import synth
    from sam6d_hip.parallel import gather_poses, shard_indices, unshard, pack_poses
    pem = _lib = W = sd = None
    if not stub:
        import sam6d_hip
        sam6d_hip.require_lib()
        from sam6d_hip import _lib, pem
        sd = synth.make_pem_weights(1)
        W = pem.PemWeights(sd, dev)

    # ------------------------------------------------------------------------------------------------ this rank's proposals
    cfg4 = args.workload == "config4"
    n_valid = None
    if cfg4:
        scene = synth.config4_inputs(B=CONFIG4_TOTAL, n_obj=CONFIG4_OBJECTS, seed=4)
        ids, n_valid = shard_indices(CONFIG4_TOTAL, rank, world)  # round-robin, padded to equal counts
        inp = {k: (scene[k][ids].contiguous() if k not in ("dense_po", "dense_fo") else scene[k]) for k in scene}
        B = len(ids)
        total_per_step = CONFIG4_TOTAL
    else:
        B = args.batch
        inp = synth.config2_inputs(B=B, seed=1 + rank)  # every rank its own shard of proposals
        total_per_step = world * B
    d = {k: v.to(dev).contiguous() for k, v in inp.items()}
    keys = ("dense_pm", "dense_fm", "dense_po", "dense_fo", "radius", "model")
    margs = [d[k] for k in keys]
    tids = d.get("template_ids")
    sync()

    opts = None if stub else pem.Options.from_env()
    use_graph = (args.graph == 1 or (args.graph < 0 and B < 16)) and not stub
    # micro-batch slices on concurrent streams: measured in round 4 (scratch/graph_probe2.py, scratch/concurrency_probe.py) to give
    # nothing on this chip -- a throughput-bound kernel keeps every CU's LDS / wave slots occupied, so another stream's latency-bound
    # launch (139 KB of LDS per workgroup) does not get on the chip beside it (L || T on two streams: 2.80 ms against 2.84 ms serial)
    mb = args.microbatch if args.microbatch > 0 else (opts.microbatch if opts else 1)
    cfg = None if stub else dict(pem.DEFAULT_CFG, microbatch=mb)
    graph = None
    graph_error = None
    if use_graph:
        try:
            graph = pem.PemGraph(W, *margs, d["rand"], cfg=cfg, template_ids=tids)
        except Exception as ex:
            graph_error = "%s: %s" % (type(ex).__name__, ex)
            sys.stderr.write("bench.py: hipGraph capture failed (%s); timing the eager launch sequence\n" % graph_error)
            use_graph = False

    def step_local():
        if stub:
            return _stub_match(d)
        if graph is not None:
            return graph.replay()
        return pem.pem_match(*margs, W, d["rand"], cfg=cfg, template_ids=tids)

    ag_events = []

    def step():
        R, t, s = step_local()
        if world == 1:
            return R, t, s
        if stub:
            return gather_poses(R, t, s, dist)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = gather_poses(R, t, s, dist)
        e1.record()
        ag_events.append((e0, e1))
        return out

    for _ in range(args.warmup):
        out = step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    ag_events.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt_local = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    ag_ms = (sum(a.elapsed_time(b) for a, b in ag_events) / max(1, len(ag_events))) if ag_events else None
    per_rank_ms = None
    if dist is not None:
        box = torch.tensor([dt, dt_local, ag_ms or 0.0], device=dev, dtype=torch.float64)
        allb = [torch.empty_like(box) for _ in range(world)]
        dist.all_gather(allb, box)
        dt = max(float(b[0]) for b in allb)
        per_rank_ms = [1e3 * float(b[1]) / args.steps for b in allb]
        ag_ms = [float(b[2]) for b in allb]

    # poses of the whole job in global proposal order (config 4: the scene's 200) -- a digest for the sharded == unsharded checks
    poses_sha = None
    if cfg4:
        g = pack_poses(*out).cpu() if world > 1 else pack_poses(*out).cpu()
        allp = unshard(g, CONFIG4_TOTAL, world) if world > 1 else g[:CONFIG4_TOTAL]
        poses_sha = hashlib.sha256(allp.contiguous().numpy().tobytes()).hexdigest()

    if stub:
        if rank == 0:
            res = {"metric": "stub", "n_gpus": world, "steps": args.steps, "workload": args.workload, "rows": int(out[0].shape[0]),
                   "poses_sha256": poses_sha, "per_rank_ms": per_rank_ms, "proposals_per_rank": B}
            line = json.dumps(res)
            print(line)
            if args.out:
                open(args.out, "w").write(line)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ------------------------------------------------------------------------------------------------ sustained clock
    sustained = None
    if args.sustained_seconds > 0:
        n = 0
        ts = time.perf_counter()
        while True:
            for _ in range(10):
                step()
            n += 10
            sync()  # (every 10 steps: ~0.1 % of the interval)
            stop = time.perf_counter() - ts >= args.sustained_seconds
            if dist is not None:
                flag = torch.tensor([1.0 if stop else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                stop = bool(flag.item() > 0)
            if stop:
                break
        sdt = time.perf_counter() - ts
        if dist is not None:
            tt = torch.tensor([sdt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            sdt = float(tt.item())
        sustained = {"ms_per_step": 1e3 * sdt / n, "value": total_per_step * n / sdt, "unit": "proposals/s", "seconds": sdt, "steps": n,
                     "burst_ms_per_step": 1e3 * dt / args.steps,
                     "note": "continuous steps right after the timed %d, same process; the chip lowers its clock under sustained load" % args.steps}

    # ---------------------------------------------------- dominant-kernel durations: HIP events in eager steps (not capturable in a graph)
    mode = _lib.load().sam6d_get_matmul_mode()
    split = mode >= 1
    paths = pem.describe_paths(W, cfg, opts)
    fused = paths["rpe_stage1_products"] is not None
    kname = "rpe_score_kernel" if fused else "geo_embed_kernel"

    def eager_local():
        return pem.pem_match(*margs, W, d["rand"], cfg=cfg, template_ids=tids)

    prof = {}
    if rank == 0:
        n_inst = max(1, min(args.steps, 5))
        pem.PROFILE_NAMES = {kname}
        pem.PROFILE = {}
        for _ in range(n_inst):
            eager_local()
        torch.cuda.synchronize()
        prof.update(pem.PROFILE)
        pem.PROFILE_NAMES = {"linattn_layer", "fine_match"}
        pem.PROFILE = {}
        for _ in range(n_inst):
            eager_local()
        torch.cuda.synchronize()
        prof.update(pem.PROFILE)
        pem.PROFILE, pem.PROFILE_NAMES = None, None

    if rank == 0:
        ev = prof.get(kname, [])
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        k_ms = sum(ms) / max(1, len(ms))
        tj, tsrc = {}, None
        for cand in (TRAFFIC_FILE, TRAFFIC_FALLBACK):
            try:
                tj = json.load(open(os.path.join(ROOT, cand)))
                tsrc = cand
                break
            except Exception:
                continue

        def _tr(prefix):
            try:
                k = [k for k in tj["kernels"] if k.startswith(prefix)]
                return tj["kernels"][k[0]]["hbm_bytes_per_launch"] if (k and B == B_PER_GPU and not cfg4) else None
            except Exception:
                return None

        def _rec(v):
            return {"hbm_bytes_per_launch": v, "source": tsrc,
                    "note": "from separate rocprofv3 --pmc passes recorded under profiles/ for a 64-cloud launch (B = 32, microbatch 1), "
                            "corrected as MI355X_MICROARCH.md prescribes (2 x FETCH_SIZE + WRITE_SIZE); NOT measured in this run"}

        clouds = 2 * ((B + mb - 1) // mb)  # one launch covers one micro-batch slice: scene + template clouds
        if fused:
            # one launch = one RPE layer over the stacked clouds of a slice (6 launches per slice and step).  `achieved` counts the
            # contraction the kernel is formulated as in fp32-equivalent flops; `frac` prices what the kernel EXECUTES -- every fp32
            # product costs np1 (stage 1: 2 where the host guard lets the cross terms of the orders >= 16 go, else 3) or 3 (stage 2)
            # fp16 MFMA products -- against the dense fp16 MFMA peak.
            flop = clouds * 197 * RPE_FLOP_PER_QUERY
            achieved = flop / (k_ms * 1e-3) / 1e12 if ev else None
            np1 = 1 if mode == 2 else int(paths["rpe_stage1_products"])
            np2 = 1 if mode == 2 else 3
            f1, f2 = 3 * 256 * 32, 4 * 256
            exec_ratio = (np1 * f1 + np2 * f2) / float(f1 + f2 + 4 * 32)  # executed fp16 MFMA flops per algorithmic flop (d part: vector)
            executed = achieved * exec_ratio if achieved else None
            ref_flop = clouds * (PROJP_FLOP_PER_CLOUD + GEO_FLOP_PER_CLOUD / 6.0)
            traffic = _tr(kname)
            if traffic is not None and mb > 1:
                traffic = traffic // mb  # recorded for a 64-cloud launch; a slice moves its share
            roofline = {"bound": "mfma",
                        "kernel": "rpe_score_kernel (v_mfma_f32_16x16x32_f16, fp16 split precision: %d MFMA products per fp32 product in stage 1, "
                                  "%d in stage 2); one RPE layer over %d clouds per launch, %d launches per step%s"
                                  % (np1, np2, clouds, 6 * mb, "" if mb == 1 else " on %d concurrent streams" % mb),
                        "achieved": executed, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": (executed / PEAK_FP16_MFMA_TFLOPS) if executed else None,
                        "traffic": traffic, "traffic_recorded": _rec(traffic),
                        "launch_ms": k_ms, "launches_timed": len(ms),
                        "timing": "HIP events around the kernel (torch's current stream = the launch stream) in %d eager steps of the same "
                                  "configuration right behind the timed region%s" % (max(1, min(args.steps, 5)),
                                                                                      "; the timed region itself replays a hipGraph, in which events cannot be recorded" if use_graph else ""),
                        "executed_fp16_mfma_gflop_per_launch": flop * exec_ratio / 1e9,
                        "executed_mfma_tflops_incl_padding": (executed * (208.0 / 197.0)) if executed else None,
                        "algorithmic_gflop_per_launch": flop / 1e9, "algorithmic_tflops": achieved,
                        "stage1_products": np1,
                        "frac_three_product_yardstick": (achieved / (PEAK_FP16_MFMA_TFLOPS / 3.0)) if achieved else None,
                        "reference_formulation_gflop_per_launch": ref_flop / 1e9,
                        "reference_formulation_tflops": (ref_flop / (k_ms * 1e-3) / 1e12) if ev else None}
        else:
            achieved = (2 * B * GEO_FLOP_PER_CLOUD) / (k_ms * 1e-3) / 1e12 if ev else None
            peak = PEAK_FP16_MFMA_TFLOPS / 3.0 if split else PEAK_FP32_MFMA_TFLOPS
            traffic = _tr("geo_embed_h3_kernel")
            roofline = {"bound": "mfma",
                        "kernel": ("geo_cheb_kernel + geo_embed_h3_kernel" if split else "geo_embed_kernel (v_mfma_f32_32x32x2_f32)")
                                  + ", 2B clouds per launch",
                        "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": (achieved / peak) if achieved else None,
                        "traffic": traffic, "traffic_recorded": _rec(traffic), "launch_ms": k_ms, "launches_timed": len(ms),
                        "algorithmic_gflop_per_launch": 2 * B * GEO_FLOP_PER_CLOUD / 1e9, "fp32_mfma_peak": PEAK_FP32_MFMA_TFLOPS}

        extra = {}
        lev = prof.get("linattn_layer", [])
        if lev:
            # the dense LinearTransformerLayer of the sparse-to-dense lift as ONE kernel (token_block_kernel<1>): clouds x 2048 tokens,
            # per token proj_q 256x256, head mix 4 x 64x64, linear 256x256, FFN 256 -> 512 -> 256; every fp32 product = 3 fp16 MFMA
            # products; algorithmic bytes = D read + D' written
            l_ms = [a.elapsed_time(b) for a, b in lev]
            ms1 = sum(l_ms) / len(l_ms)
            tok = clouds * 2048
            fl = tok * 2.0 * (2 * 256 * 256 + 256 * 64 + 2 * 256 * 512)
            ach = fl / (ms1 * 1e-3) / 1e12
            nprod = 1.0 if mode == 2 else 3.0
            tr = _tr("token_block_kernel<1")
            extra["roofline_dense_layer"] = {
                "bound": "mfma", "kernel": "token_block_kernel<1> (sam6d_linattn_layer): one dense linear-attention layer over %d clouds, "
                "%d launches per step" % (clouds, 3 * mb),
                "achieved": ach * nprod, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach * nprod / PEAK_FP16_MFMA_TFLOPS,
                "traffic": (tr // mb) if tr else None, "traffic_recorded": _rec((tr // mb) if tr else None), "launch_ms": ms1,
                "launches_timed": len(l_ms), "algorithmic_gflop_per_launch": fl / 1e9, "algorithmic_tflops": ach,
                "algorithmic_mb_per_launch": 2 * tok * 256 * 4 / 1e6, "hbm_gbs_algorithmic": 2 * tok * 256 * 4 / 1e9 / (ms1 * 1e-3)}
        fev = prof.get("fine_match", [])
        if fev:
            # similarity + soft assignment of the fine stage (7 launches of finematch.hip): an HBM stream -- features read and written as
            # fp16 hi/lo, E = exp(att - c) (b x 2049 x 2052 floats) written once and read twice
            f_ms = [a.elapsed_time(b) for a, b in fev]
            ms1 = sum(f_ms) / len(f_ms)
            bs = (B + mb - 1) // mb
            Eb = bs * 2049 * 2052 * 4
            feat = 2 * bs * 2049 * 256 * 4
            alg = feat + feat + feat + Eb + feat + Eb + Eb  # prep r + w, sim operands, E write, bg rows, labels read, assign read
            gbs = alg / 1e9 / (ms1 * 1e-3)
            ft = (tj.get("fine_match_total_bytes") if (B == B_PER_GPU and not cfg4) else None) if isinstance(tj, dict) else None
            extra["roofline_fine_match"] = {
                "bound": "hbm", "kernel": "sam6d_fine_match (fm_sim / fm_bg / fm_labels / fm_assign + merges), once per slice and step",
                "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                "traffic": (ft // mb) if ft else None, "traffic_recorded": _rec((ft // mb) if ft else None),
                "launch_ms": ms1, "launches_timed": len(f_ms), "algorithmic_mb_per_launch": alg / 1e6}

        if cfg4:
            metric = "proposals/sec through PEM match+SVD (LM-O-style scene, 200 proposals of 8 objects round-robin over the GPUs); pose Δ vs CPU ref"
            workload = ("ONE scene of %d proposals of %d objects (SURVEY 8d config 4), proposal b on rank b %% %d: %d per GPU; unique "
                        "templates + template_ids; 2048 scene + 2048 model pts, 1024 CAD pts, random-init weights"
                        % (CONFIG4_TOTAL, CONFIG4_OBJECTS, world, B))
        else:
            metric = "proposals/sec through PEM match+SVD (B=32, 2048 pts); pose Δ vs CPU ref"
            workload = ("PEM batch=%d proposals/GPU, 2048 scene + 2048 model pts, 1024 CAD pts, random-init weights (SURVEY 8d config 2)" % B)
        res = {
            "metric": metric,
            "value": total_per_step * args.steps / dt, "unit": "proposals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong" if cfg4 else "weak", "vs_baseline": None,
            "dtype": ("f16 (EXPERIMENTAL single-product fp16 MFMA, pose parity unpinned)" if mode == 2 else
                      "f32 (fp16x3 split-precision MFMA, fp32 accumulate)" if split else "f32"), "data": "synthetic",
            "config": {"workload": workload, "proposals_per_gpu": B,
                       "parallelism": "proposal-sharded x%d, RCCL all-gather of 13 floats/proposal" % world,
                       "rccl_world_size": (dist.get_world_size() if dist is not None else 1),
                       "launch": ("hipGraph replay (one graph launch per step), %d micro-batch slice(s) on concurrent streams" % mb) if use_graph
                                 else "eager: one ctypes launch per kernel, %d micro-batch slice(s)" % mb,
                       "matmul": ("fp16 single-product MFMA, fp32 accumulate (~1e-3 rel.; EXPERIMENTAL: poses are not preserved on random-init "
                                  "weights, DESIGN 4 'Mode 2' -- config 5 is measured in the default mode, key `config5`)" if mode == 2 else
                                  "fp16x3 split-precision MFMA, fp32 accumulate (~1e-6 rel.)" if split else "exact fp32 MFMA")},
            "paths": paths,
            "roofline": roofline,
        }
        if graph_error:
            res["config"]["graph_error"] = graph_error
        if sustained:
            res["sustained"] = sustained
        if per_rank_ms is not None:
            res["per_rank_ms"] = per_rank_ms
            res["allgather_ms"] = ag_ms
        if poses_sha:
            res["poses_sha256"] = poses_sha
        res.update(extra)
        extras = world == 1 and not args.no_extras
        if extras:
            try:
                res["latency"] = bench_latency(dev, W, args.graph != 0)
            except Exception as e:
                res["latency"] = {"error": "%s: %s" % (type(e).__name__, e)}
            if not cfg4:
                try:
                    res["config4"] = bench_config4_single(dev, W, args.graph == 1, 1)
                except Exception as e:
                    res["config4"] = {"error": "%s: %s" % (type(e).__name__, e)}
                try:
                    res["fallbacks"] = bench_fallbacks(dev, sd, d)
                except Exception as e:
                    res["fallbacks"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_ism:
            try:
                res["ism_config3"] = bench_ism(dev)
            except Exception as e:  # the headline line must not depend on the side measurement
                res["ism_config3"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and not args.no_config5 and mode == 1:
            try:
                res["config5"] = bench_config5(dev, W)
            except Exception as e:
                res["config5"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if args.cpu_proposals > 0 and world == 1:
            # host cores for the baseline: the GPU box gives a 1-GPU job a share of 16 cores (more threads only
            # oversubscribe the shared host: 256 threads ran the same port 20x slower)
            threads = min(len(os.sched_getaffinity(0)), 16)
            v, cdt, (cR, ct, cs), cinp = cpu_baseline(sd, args.cpu_proposals, threads)
            res["cpu_baseline"] = {"value": v, "unit": "proposals/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
                                   "chunk": CPU_CHUNK,
                                   "sample": "%d proposals of the config-2 generator (seed 1), %d per call, oracle/pem_oracle.py "
                                             "with the reference's dense sampling compare; %.1f s" % (args.cpu_proposals, CPU_CHUNK, cdt)}
            # pose delta of the GPU path vs the CPU port on the same proposals
            n = args.cpu_proposals
            g = {k: v[:n].to(dev).contiguous() for k, v in cinp.items()}
            R, t, s = pem.pem_match(g["dense_pm"], g["dense_fm"], g["dense_po"], g["dense_fo"], g["radius"], g["model"], W, g["rand"])
            res["pose_delta_vs_cpu"] = {"max_abs_dR": float((R.cpu() - cR).abs().max()), "max_abs_dt": float((t.cpu() - ct).abs().max()),
                                        "max_abs_dscore": float((s.cpu() - cs).abs().max()), "proposals": n}
            res["speedup_vs_cpu_baseline"] = res["value"] / v
        line = json.dumps(res)
        print(line)
        if args.out:
            open(args.out, "w").write(line)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
