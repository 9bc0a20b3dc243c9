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


@pytest.mark.parametrize("skip_dropped", [False, True])
def test_two_rank_step_keeps_replicas_identical(skip_dropped):
    """skip_dropped: the same with TransformerEncoder.skip_dropped_branches -- every rank packs ITS kept samples (other row counts,
    other launch sizes per rank and layer); the collectives and the bucket order must not notice."""
    env = dict(os.environ, ONEPEACE_DIST_BACKEND="gloo", ONEPEACE_SINGLE_DEVICE_DEBUG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "8", "--layers", "3", "--check-replicas", "--no-profile"] + (["--skip-dropped"] if skip_dropped else [])
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:  # the ranks' own tracebacks come before torchrun's summary: keep the whole log
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "dist_test_stderr.txt"), "w").write(r.stderr)
    assert r.returncode == 0, r.stderr[-6000:]
    assert "replica check: IDENTICAL" in r.stderr, r.stderr[-2000:]
    assert '"n_gpus": 2' in r.stdout


def _run_bench(nproc, extra_env, extra_args=()):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "3", "--warmup", "1",
           "--batch", "8", "--layers", "3", "--no-cpu-baseline", *extra_args]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "dist_test_stderr.txt"), "w").write(r.stderr)
    assert r.returncode == 0, r.stderr[-6000:]
    return r


def test_rccl_collectives_path_at_world_one():
    """One rank, backend nccl (= RCCL), collectives forced on (ONEPEACE_FORCE_COLLECTIVES): the broadcast, the fused [3, b, H]
    all-gather and the bucketed gradient all-reduce run through RCCL on the device and the step still trains."""
    import json
    r = _run_bench(1, {"ONEPEACE_FORCE_COLLECTIVES": "1"})
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["config"]["collectives"].startswith("nccl")
    assert line["config"]["final_loss"] == line["config"]["final_loss"]  # not NaN


def test_one_tile_launch_rule_of_multi_gpu_runs_gives_the_same_bits_over_rccl():
    """The launch rule the first multi-GPU run takes (distributed.share_cus_with_collectives: every NT GEMM as one tile per workgroup,
    tune sched 7, selected by init_distributed for world > 1 over RCCL) against the single-GPU rule (persistent workgroups), both over
    RCCL at world 1 with the collectives forced on: the two rules launch bit-identical GEMM tiles (tests/test_ops_gpu.py), so the same
    training steps must give the same loss at every step and the same parameters after the last optimiser step (checksums to 1e-9
    relative: the relative-position table gradient is scattered with fp32 atomics, the one run-to-run freedom of the step).
    (the rule had never run under a test.)"""
    import json
    lines = {}
    for rule in ("0", "1"):
        r = _run_bench(1, {"ONEPEACE_FORCE_COLLECTIVES": "1", "ONEPEACE_SHARE_CUS": rule}, ("--loss-curve", "--no-skip-leg", "--no-power-probe"))
        lines[rule] = json.loads(r.stdout.strip().splitlines()[-1])["config"]
        assert ("one tile per workgroup" in r.stderr) == (rule == "1"), r.stderr[-1500:]  # the switch is logged when it is applied
    assert lines["0"]["nt_gemm_launch_rule"].startswith("persistent") and lines["1"]["nt_gemm_launch_rule"].startswith("one tile")
    assert lines["0"]["collectives"].startswith("nccl") and lines["1"]["collectives"].startswith("nccl")
    assert lines["0"]["loss_curve"] == lines["1"]["loss_curve"] and len(lines["0"]["loss_curve"]) == 3
    assert lines["0"]["final_loss"] == lines["1"]["final_loss"]
    for a, b in zip(lines["0"]["param_checksum"], lines["1"]["param_checksum"]):
        assert abs(a - b) <= 1e-9 * abs(a), (lines["0"]["param_checksum"], lines["1"]["param_checksum"])


def test_two_rank_step_on_rccl_keeps_replicas_identical():
    """The data-parallel step over RCCL with two ranks on two devices (skipped on a single-GPU box): different data per rank,
    bit-identical replicas afterwards, every gradient bucket all-reduced during backward (the 3-pass shared-attention case),
    and the overlap report present in the bench line (legacy_distributed_data_parallel.py:76-165)."""
    import json

    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the driver's multi-GPU tier runs it)")
    r = _run_bench(2, {}, ("--check-replicas",))
    assert "replica check: IDENTICAL" in r.stderr, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    rep = line["config"]["grad_allreduce_overlap"]
    # (the very first step learns which parameters are unused -- the mask embeddings -- so its buckets may wait for finish())
    assert line["n_gpus"] == 2 and rep["launched_in_finish"] <= rep["buckets"]
    assert rep["launched_in_backward"] >= rep["buckets"] * (rep["steps"] - 1)
    assert "exposed_allreduce_ms_per_step" in rep
