"""`python bench.py --gpus N` must START N ranks (the driver types exactly that; round 4's bench.py parsed --gpus and never read it).
CPU: the launcher re-executes bench.py under torch.distributed.run; --dry-run-launch forms the process group (gloo here, nccl = RCCL
with GPUs), all-reduces a one per rank and prints ONE line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_launch_command_is_one_rank_per_gpu_on_localhost():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "3"], port=1234)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "1234"
    assert cmd[-5:] == [os.path.abspath(BENCH), "--gpus", "8", "--steps", "3"]


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-launch"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_reduced"] == 2 and line["requested_gpus"] == 2
    assert "rank 0 of 2 up" in r.stderr and "rank 1 of 2 up" in r.stderr


def test_gpus_2_without_gpus_fails_in_both_ranks_not_in_one():
    env = _env()
    env["SEAL_BENCH_FAIL_LINGER_S"] = "15"      # (the launcher kills the slower rank when the faster one exits: both must get to say why)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "needs a GPU (rank 0 of 2)" in r.stderr and "needs a GPU (rank 1 of 2)" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_a_launcher_that_set_the_world_is_respected():
    # under the driver's own `python -m torch.distributed.run ... bench.py --gpus 2` RANK/WORLD_SIZE exist: no second launcher
    env = _env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-launch"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "launching" not in r.stderr
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_cpu_smoke_two_ranks_gather_the_single_rank_topk():
    """bench.py's REAL N > 1 path -- `python bench.py --gpus 2` becomes the launcher, two gloo ranks shard the queries, run the product's
    SEALSearcher.batch_search (tiny CPU searcher, oracle-answered index), all-gather the top-k, and rank 0 prints ONE line -- must gather
    bit for bit what one rank computes over all the queries.  So the first run on 8 GPUs is not also the first run of this plumbing."""
    outs = {}
    for n in (1, 2):
        r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--cpu-smoke", "--topk", "10"], env=_env(), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        outs[n] = json.loads(lines[0])
    one, two = outs[1], outs[2]
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["requested_gpus"] == 2 and two["cpu_smoke"] is True
    assert two["shards"][0][0] == 0 and two["shards"][0][1] == two["shards"][1][0] and two["shards"][1][1] == two["queries"]
    assert two["topk_shape"] == one["topk_shape"] == [one["queries"], 10, 2]
    assert two["topk_hex"] == one["topk_hex"] and two["topk_sha256"] == one["topk_sha256"]
    import numpy as np
    top = np.frombuffer(bytes.fromhex(two["topk_hex"]), dtype=np.float64).reshape(two["topk_shape"])
    assert (top[:, 0, 0] >= 0).all()                      # every query found documents
    assert two["per_rank_queries_per_s_min"] > 0
