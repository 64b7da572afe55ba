"""The N > 1 path of bench.py on the GPU (-m gpu): two ranks launched with torch.distributed.run as the driver launches
them, both mapped onto the one GPU of the box (AG_BENCH_SHARE_GPU=1) with gloo in RCCL's place (RCCL refuses two ranks on
one device) - shards, per-rank rollouts, the MAX all-reduce of the two batch-global scalars and the all-gather of the
rewards all run.  The gathered reward vector must equal the one-rank run's bit for bit (candidates are independent; no
float atomics).  Children only: this process never re-executes itself.  RCCL itself executes the same collectives with a
world of ONE rank (test below); it has still never seen more than one rank (no multi-GPU node has been available):
DESIGN.md section 6."""
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


_ONE = {}


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                             "AG_BENCH_FORCE_DIST", "AG_BENCH_BACKEND", "AG_BENCH_SHARE_GPU")}


def _one_rank_line():
    """the plain one-rank bench line (no process group), run once per test session in a child process"""
    if "line" not in _ONE:
        one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS, env=_clean_env(), cwd=ROOT,
                             capture_output=True, text=True, timeout=900)
        assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
        _ONE["line"] = _line(one.stdout)
    return _ONE["line"]


def test_one_rank_runs_the_rccl_collectives_and_equals_the_plain_run():
    """AG_BENCH_FORCE_DIST=1: `bench.py --gpus 1` initialises the nccl (= RCCL) backend with a world of one (communicator
    created on device_id) and goes through sharded_candidate_rewards -> all_gather_costs -> dist.all_gather_into_tensor and both
    MAX all-reduces of running_cost / cloth_penalty ON RCCL - the calls a multi-GPU run makes, on the hardware that exists.
    Fresh child process (never a re-exec of one that touched the GPU).  Same reward bits as the plain run."""
    l1 = _one_rank_line()
    env = dict(_clean_env(), AG_BENCH_FORCE_DIST="1")
    forced = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + ARGS, env=env, cwd=ROOT,
                            capture_output=True, text=True, timeout=900)
    assert forced.returncode == 0, forced.stdout[-2000:] + forced.stderr[-4000:]
    lf = _line(forced.stdout)
    assert "multi_gpu" not in l1
    mg = lf["multi_gpu"]
    assert mg["backend"] == "nccl" and mg["candidates_per_rank"] == [64] and mg["exchange_us_per_step"] > 0, mg
    assert lf["n_gpus"] == 1 and lf["reward_sha256"] == l1["reward_sha256"], (lf["reward_sha256"], l1["reward_sha256"])
    print(f"one-rank RCCL exchange (all-reduce MAX of two scalars + all-gather of 64 rewards): {mg['exchange_us_per_step']:.1f} us per step")


def test_two_ranks_on_one_gpu_equal_one_rank_bitwise():
    env = _clean_env()
    l1 = _one_rank_line()
    env2 = dict(env, AG_BENCH_SHARE_GPU="1", AG_BENCH_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, env=env2, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    l2 = _line(two.stdout)
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2
    assert l2["config"]["candidates"] == 64 and "sharded over 2 GPU(s)" in l2["config"]["parallelism"]
    assert l1["reward_sha256"] == l2["reward_sha256"], (l1["reward_sha256"], l2["reward_sha256"])
    assert l2["value"] > 0 and l2["scaling"] == "strong"
    mg = l2["multi_gpu"]                                                 # per-rank times and the exchange's latency, for a real SCALE run
    assert mg["candidates_per_rank"] == [32, 32] and len(mg["per_rank_ms_per_step"]) == 2 and mg["exchange_us_per_step"] > 0


def test_bare_command_launches_its_own_two_ranks():
    """`python3 bench.py --gpus 2` with NO launcher in front (the shape of the only command the driver has issued so far): the
    parent starts torch.distributed.run as a child before anything touches the GPU; one line comes out, n_gpus 2, world size as
    torch.distributed sees it, same reward bits as one rank.  (CPU plumbing of the same entry: tests/test_bench_launch.py.)"""
    l1 = _one_rank_line()
    env2 = dict(_clean_env(), AG_BENCH_SHARE_GPU="1", AG_BENCH_BACKEND="gloo")
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + ARGS, env=env2, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    l2 = _line(two.stdout)
    assert l2["n_gpus"] == 2 and l2["reward_sha256"] == l1["reward_sha256"]
    mg = l2["multi_gpu"]
    assert mg["world_size"] == 2 and mg["launched_by"] == "bench.py" and mg["backend"] == "gloo"
    assert [r["rank"] for r in mg["ranks"]] == [0, 1] and len({r["pid"] for r in mg["ranks"]}) == 2
    assert mg["distinct_devices"] == 1                                    # the rehearsal shares the one GPU; a real node reports N
    assert len(mg["per_rank_ms_per_step"]) == 2 and mg["exchange_us_per_step"] > 0


def test_work_balanced_shards_of_the_planner_workload_on_two_ranks():
    """The shipped planner's pushes (uniform over the action box: most never reach the rope) sharded over two ranks by WORK -
    forwards left per candidate from adaptigraph_amd.rollout_work, identical on every rank, no exchange - on the one GPU of the
    box (gloo): the gathered reward vector equals the count-balanced one's and the one-rank evaluation's bit for bit, the two
    shards hold different numbers of candidates and about the same work."""
    env = dict(_clean_env(), AG_SHARD_CANDIDATES="3000")
    tool = os.path.join(ROOT, "tools", "two_rank_planner_shards.py")
    one = subprocess.run([sys.executable, tool], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), tool],
                         env=dict(env, AG_BENCH_SHARE_GPU="1"), cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
    l2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
    assert l1["world"] == 1 and l2["world"] == 2
    assert l2["reward_sha256_work_balanced"] == l2["reward_sha256_count_balanced"] == l1["reward_sha256_work_balanced"]
    assert 0.5 < l2["never_touch_fraction"] < 0.95
    wb, cb = l2["work_balanced"], l2["count_balanced"]
    f = wb["forwards_per_rank"]
    assert abs(f[0] - f[1]) <= 0.1 * max(f) + 16, f                       # cut by work: within 10 % (+ one candidate's repeats)
    n = [r["n"] for r in wb["per_rank"]]
    assert sum(n) == 3000 and n == [b - a for a, b in wb["bounds"]]
    # what each rank executed = its candidates' forwards left (+ nothing for the kept base rollout on the second pass)
    assert [r["executed"] for r in wb["per_rank"]] == f, (wb["per_rank"], f)
    print(f"2 ranks, {l2['never_touch_fraction']:.0%} never touch: work-balanced forwards {f} (candidates {n}), count-balanced "
          f"{cb['forwards_per_rank']}")


def test_the_unchanged_planner_loop_dealt_to_two_ranks_equals_one_rank():
    """The reference's own chunk loop (plan.py:210, 241-247) on the shipped rope configuration, planner_config['group'] set, two
    ranks on the one GPU of the box (gloo): the merged result - winning action, its rollout, its reward - equals the one-rank loop's
    bit for bit, every rank holds it, the generators end in the same state, each rank rolled out every second call only.  Also
    with random_interact.py's n_update_iter 5 (the in-call best-so-far selection stays on the device)."""
    tool = os.path.join(ROOT, "tools", "two_rank_planner_loop.py")
    for upd, chunks in ((1, 12), (5, 3)):
        env = dict(_clean_env(), AG_LOOP_CHUNKS=str(chunks), AG_LOOP_REPS="1", AG_LOOP_UPDATE_ITER=str(upd))
        one = subprocess.run([sys.executable, tool], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
        l1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][-1])
        two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), tool],
                             env=dict(env, AG_BENCH_SHARE_GPU="1"), cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-4000:]
        l2 = json.loads([ln for ln in two.stdout.splitlines() if ln.startswith("{")][-1])
        assert l1["world"] == 1 and l2["world"] == 2 and l2["n_update_iter"] == upd
        assert l2["same_result_on_every_rank"] and l2["generators_in_step"]
        assert l2["result_sha256"] == l1["result_sha256"] and l2["generator_sha256"] == l1["generator_sha256"], (l1, l2)
        assert l2["calls_owned_per_rank"] == [(chunks + 1) // 2, chunks // 2]
        print(f"n_update_iter {upd}, {chunks} chunks: planner call {l1['ms_per_planner_call']:.1f} ms on one rank, "
              f"{l2['ms_per_planner_call']:.1f} ms on two ranks sharing the GPU")


def test_bare_command_with_three_ranks_and_uneven_shards():
    """`python3 bench.py --gpus 3` (no launcher), 67 candidates = shards of 23 / 22 / 22, three ranks on the one GPU of the box (gloo).
    Three, not more: the box allows six processes on its GPU at once, and this pytest process, the launcher's agent and the ranks all
    count.  Reward SHA-256 equal to the one-rank run of the same batch."""
    args = [a if a != "64" else "67" for a in ARGS]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, env=_clean_env(), cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-4000:]
    l1 = _line(one.stdout)
    three = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3"] + args,
                           env=dict(_clean_env(), AG_BENCH_SHARE_GPU="1", AG_BENCH_BACKEND="gloo"), cwd=ROOT,
                           capture_output=True, text=True, timeout=900)
    assert three.returncode == 0, three.stdout[-2000:] + three.stderr[-4000:]
    l3 = _line(three.stdout)
    mg = l3["multi_gpu"]
    assert l3["n_gpus"] == 3 and mg["world_size"] == 3 and mg["candidates_per_rank"] == [23, 22, 22]
    assert l3["reward_sha256"] == l1["reward_sha256"] and l1["config"]["candidates"] == 67
