"""Data-parallel step on the HIP path with two ranks (one device, gloo transport for the collectives): every rank sees
different data, so the replicas only stay bit-identical if every gradient -- incl. the ones the fused layers accumulate in
place in the flat buffer -- was all-reduced after its last contribution, the clip norm was taken over the reduced buffer, and
the optimiser applied the same update everywhere (legacy_distributed_data_parallel.py:76-165, trainer.py:917-935)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_step_keeps_replicas_identical():
    env = dict(os.environ, ONEPEACE_DIST_BACKEND="gloo", ONEPEACE_SINGLE_DEVICE_DEBUG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--layers", "3", "--check-replicas", "--no-profile"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:  # the ranks' own tracebacks come before torchrun's summary: keep the whole log
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "dist_test_stderr.txt"), "w").write(r.stderr)
    assert r.returncode == 0, r.stderr[-6000:]
    assert "replica check: IDENTICAL" in r.stderr, r.stderr[-2000:]
    assert '"n_gpus": 2' in r.stdout
