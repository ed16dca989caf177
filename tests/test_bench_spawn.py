"""bench.py --gpus N from a plain invocation: the parent starts N ranks itself (no torchrun), and a job whose size differs
from --gpus fails instead of degrading (the metric is reported at 1, 2, 4 and 8 GPUs)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], env=env, capture_output=True, text=True, timeout=300)


def test_plain_invocation_spawns_n_ranks():
    r = _run(["--gpus", "3", "--spawn-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout     # ONE JSON line: gloo's connection banners go to stderr
    line = json.loads(r.stdout)
    assert line["n_gpus"] == 3 and line["dist_world_size"] == 3
    assert line["rank_sum"] == 6.0                      # every rank took part in the collective: 1 + 2 + 3
    assert line["launcher"] == "bench.py"


def test_external_launcher_is_respected():
    r = _run(["--gpus", "1", "--spawn-check"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["launcher"] == "external"


def test_world_size_mismatch_fails():
    r = _run(["--gpus", "4", "--spawn-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr
    r = _run(["--gpus", "1", "--spawn-check"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0


def test_a_failing_rank_fails_the_job():
    # rank 1 exits before the rendezvous: the launcher must take the other rank down (it would wait in the collective) and fail
    r = _run(["--gpus", "2", "--spawn-check"], {"B4D_BENCH_TEST_FAIL_RANK": "1"})
    assert r.returncode != 0 and "ranks failed" in r.stderr
    assert "spawn_check" not in r.stdout
