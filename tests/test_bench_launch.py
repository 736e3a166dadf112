"""CPU: `python bench.py --gpus N` with no launcher on the command line starts its N ranks itself (bench.py: launch_ranks) and relays
rank 0's JSON line - the reference's one-worker-per-GPU spawn (saber/utils/parallelization.py:137-151, 339-343).  The ranks only
rendezvous over gloo here (SABER_AMD_BENCH_SPAWN_ONLY=1): no GPU is touched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(SABER_AMD_BENCH_SPAWN_ONLY="1", **(extra_env or {}))
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"],
                          env=env, capture_output=True, text=True, timeout=300)


def test_bench_starts_its_own_ranks():
    r = _run(2)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]       # (gloo's C++ side prints a connection note on stdout; RCCL does not)
    assert len(lines) == 1, r.stdout                     # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3.0
    assert out["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]        # every rank sees the parent's command line


def test_bench_launcher_reports_a_failed_rank():
    import time
    t0 = time.time()
    r = _run(2, {"SABER_AMD_BENCH_SPAWN_FAIL_RANK": "1"})
    assert r.returncode != 0
    # the surviving rank sits in a 20-s rendezvous (spawn_selftest); the launcher ends it instead of waiting for it (a real rank would
    # wait in an RCCL collective with no timeout at all: seen on a one-GPU box, 420 s until the box's watchdog)
    assert time.time() - t0 < 15
