"""The N > 1 path of bench.py on the GPU (-m gpu): two ranks launched with torch.distributed.run as the driver launches
them, both mapped onto the one GPU of the box (AG_BENCH_SHARE_GPU=1) with gloo in RCCL's place (RCCL refuses two ranks on
one device) - shards, per-rank rollouts, the MAX all-reduce of the two batch-global scalars and the all-gather of the
rewards all run.  The gathered reward vector must equal the one-rank run's bit for bit (candidates are independent; no
float atomics).  Children only: this process never re-executes itself.  RCCL itself has still never seen more than one
rank (no multi-GPU node has been available): DESIGN.md section 6."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--candidates", "64", "--side", "20", "--steps", "1", "--warmup", "0", "--no-bf16x3", "--no-mpc-iter",
        "--no-cpu-baseline", "--no-kernel-profile"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _line(out):
    rows = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(rows) == 1, out[-3000:]
    return json.loads(rows[0])


def test_two_ranks_on_one_gpu_equal_one_rank_bitwise():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    env2 = dict(env, AG_BENCH_SHARE_GPU="1", AG_BENCH_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, env=env2, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    l1, l2 = _line(one.stdout), _line(two.stdout)
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2
    assert l2["config"]["candidates"] == 64 and "sharded over 2 GPU(s)" in l2["config"]["parallelism"]
    assert l1["reward_sha256"] == l2["reward_sha256"], (l1["reward_sha256"], l2["reward_sha256"])
    assert l2["value"] > 0 and l2["scaling"] == "strong"
    mg = l2["multi_gpu"]                                                 # per-rank times and the exchange's latency, for a real SCALE run
    assert mg["candidates_per_rank"] == [32, 32] and len(mg["per_rank_ms_per_step"]) == 2 and mg["exchange_us_per_step"] > 0
