"""`python bench.py --gpus N` must start N ranks itself (reference: tools/train_net.py:80-90 `launch(main, num_gpus)`)
and report the size of the group it really ran on.  CPU: the launcher path with the gloo backend and no GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    return env


def test_gpus_flag_spawns_that_many_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--backend", "gloo"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 0b11


def test_external_launcher_must_agree_with_the_flag():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-check", "--backend", "gloo"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "must agree" in p.stderr


def test_a_dead_rank_fails_the_launch():
    env = dict(_env(), WSOVOD_BENCH_FAIL_RANK="1")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check", "--backend", "gloo"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert p.stdout.strip() == ""


def test_driver_launch_form_under_torch_distributed_run():
    """The driver's N > 1 form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- bench.py must take RANK / WORLD_SIZE from the launcher and not spawn again."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--launch-check",
                        "--backend", "gloo"], env=_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [x for x in p.stdout.splitlines() if x.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 0b11
