"""`bench.py --gpus N` started PLAIN (no torch.distributed.run around it, no RANK in the environment) must start the N ranks
itself: one process per GPU on the loopback address.  Checked here without a GPU through --rendezvous-only (the ranks form the
gloo group and count themselves); the GPU box runs the real two-rank step in tests/test_gpu_round3.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain(*flags, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, env=env,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_plain_launch_starts_two_ranks():
    rec = _plain("--gpus", "2", "--backend", "gloo", "--gpus-shared", "--rendezvous-only", "--steps", "3", "--warmup", "1")
    assert rec["n_gpus"] == 2 and rec["n_ranks_seen"] == 2 and rec["requested_gpus"] == 2


def test_plain_launch_single_rank_stays_in_process():
    rec = _plain("--gpus", "1", "--rendezvous-only")
    assert rec["n_gpus"] == 1 and rec["n_ranks_seen"] == 1
