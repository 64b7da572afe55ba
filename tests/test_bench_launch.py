"""`python bench.py --gpus N` with no launcher in front must start its own ranks (CPU check, no GPU): the parent - which has
imported neither torch nor the engine - runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
... bench.py <same args>` as a child process, rank 0's JSON line comes out of the parent's stdout, the parent's exit code is
non-zero when a rank failed; under an existing launcher (RANK / WORLD_SIZE set) nothing is launched.  `--plumbing-only` stops
every rank before anything touches a GPU and prints what the launch gave it.  The GPU form of the same entry (two ranks on the
one GPU of the box, real rollouts, reward SHA-256 equal to one rank) is tests/test_gpu_two_ranks.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def _line(out):
    rows = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(rows) == 1, out[-3000:]
    return json.loads(rows[0])


def test_launch_command_is_the_drivers_line():
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "2"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[3:10] == ["--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", "29511"]
    assert cmd[10] == BENCH and cmd[11:] == ["--gpus", "4", "--steps", "2"]


def test_no_launch_for_one_gpu_or_under_a_launcher(monkeypatch):
    import bench
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.launch_ranks_if_needed(["--gpus", "1"]) is None
    assert bench.launch_ranks_if_needed([]) is None
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench.launch_ranks_if_needed(["--gpus", "8"]) is None


def test_bare_command_starts_its_own_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--plumbing-only", "ok"], env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = _line(p.stdout)
    assert line["n_gpus"] == 2 and line["launched_by"] == "bench.py"
    seen = line["plumbing"]
    assert [s["rank"] for s in seen] == [0, 1] and [s["local_rank"] for s in seen] == [0, 1]
    assert all(s["world_env"] == 2 and s["world_dist"] == 2 for s in seen)
    assert len({s["pid"] for s in seen}) == 2 and os.getpid() not in {s["pid"] for s in seen}
    assert all(s["argv"] == ["--gpus", "2", "--steps", "2", "--plumbing-only", "ok"] for s in seen)      # same args, every rank
    assert "without a launcher" in p.stderr


def test_a_failing_rank_fails_the_command():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "fail-rank-1"], env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0, p.stdout[-2000:]


def test_torchrun_wrapped_form_is_unchanged():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--plumbing-only", "ok"], env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = _line(p.stdout)
    assert line["n_gpus"] == 2 and line["launched_by"] == "external launcher"
    assert "without a launcher" not in p.stderr


def test_bare_command_with_eight_ranks():
    """the node's size: `python bench.py --gpus 8` alone starts eight ranks that meet (gloo, no GPU: --plumbing-only)"""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--plumbing-only", "ok"], env=_env(), cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = _line(p.stdout)
    assert line["n_gpus"] == 8 and [s["local_rank"] for s in line["plumbing"]] == list(range(8))
    assert all(s["world_dist"] == 8 for s in line["plumbing"]) and len({s["pid"] for s in line["plumbing"]}) == 8
