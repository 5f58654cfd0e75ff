"""bench.py --gpus N without a launcher: bench.launch_ranks starts N fresh ranks itself (before any GPU call) with the environment
torch.distributed.run would provide.  Exercised here on CPU: the children are tests/helpers/rank_probe.py (gloo, the same barrier +
gather_poses sequence bench.py runs around its timed region)."""
import json
import os
import subprocess
import sys

from tests._util import ROOT

PROBE = os.path.join(ROOT, "tests", "helpers", "rank_probe.py")


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_launch_two_ranks_gather(tmp_path):
    out = tmp_path / "line.json"
    rc = _bench().launch_ranks(2, ["--gpus", "2", "--out", str(out)], script=PROBE, check_devices=False)
    assert rc == 0
    line = json.loads(out.read_text())
    assert line == {"n_gpus": 2, "world_size": 2, "rows": 8, "ok": True}


def test_launch_failed_rank_stops_the_job():
    rc = _bench().launch_ranks(2, ["--gpus", "2", "--fail-rank", "1"], script=PROBE, check_devices=False)
    assert rc == 3


def test_bench_refuses_fewer_gpus_than_asked():
    """On a box with fewer devices than --gpus the bench must fail loudly, never print an n_gpus=1 line."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert p.returncode == 2 and "needs 2 visible GPUs" in p.stderr and "n_gpus" not in p.stdout


def test_bench_rejects_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert p.returncode == 2 and "WORLD_SIZE=4" in p.stderr


def test_config4_two_gloo_ranks_equal_the_single_process_run(tmp_path):
    """bench.py --workload config4 (ONE scene of 200 proposals of 8 objects, proposal b on rank b % N, strong scaling) through
    launch_ranks with two gloo ranks: the poses gathered over the ranks and put back into global proposal order are bit for bit the
    single-process result.  The per-proposal compute is bench.py's --stub-compute stand-in (a deterministic function of every input
    of a proposal and of its template: the GPU kernels themselves are covered by the -m gpu tests, sharded == unsharded included);
    everything else -- scene generation, sharding with template_ids, the collective, unshard -- is the code the GPU run executes."""
    one, two = tmp_path / "one.json", tmp_path / "two.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "config4", "--stub-compute", "--steps", "1",
                        "--warmup", "0", "--out", str(one)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    rc = _bench().launch_ranks(2, ["--gpus", "2", "--workload", "config4", "--stub-compute", "--backend", "gloo", "--steps", "1", "--warmup", "0",
                                   "--out", str(two)], check_devices=False)
    assert rc == 0
    a, b = json.loads(one.read_text()), json.loads(two.read_text())
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and a["proposals_per_rank"] == 200 and b["proposals_per_rank"] == 100
    assert b["rows"] == 200 and len(b["per_rank_ms"]) == 2
    assert a["poses_sha256"] == b["poses_sha256"], "sharded poses differ from the single-process result"
