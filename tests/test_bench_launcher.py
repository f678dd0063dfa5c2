"""`python bench.py --gpus N` must become N ranks by itself (the driver's N > 1 command is torchrun, but a
plain invocation used to run ONE rank and print n_gpus 1).  --launch-check is the rendezvous alone (gloo, no GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 3


def test_gpus_1_goes_through_the_same_launcher():
    r = _run(["--gpus", "1", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


def test_world_size_must_match_gpus():
    # started as a rank (the torchrun form) with a world that is not --gpus: refuse instead of printing a wrong n_gpus
    r = _run(["--gpus", "2", "--launch-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                  "MASTER_PORT": "29999"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_a_dying_rank_fails_the_launch():
    r = _run(["--gpus", "2", "--launch-check", "--shards", "-7"], {"ZMI_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
