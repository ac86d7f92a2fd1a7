"""The RCCL gradient path on one GPU: a 1-rank NCCL group with the collectives forced on must reproduce the
collective-free step (side stream, stage events, bucket ranges).  Runs in a child process (own process group)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_single_rank_allreduce_matches_plain_step():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_dist_single.py")], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST SINGLE OK" in r.stdout
