"""World-size-2 checks of the data-parallel path on CPU (gloo): fused all-gather ordering / target offsets of the
contrastive criteria and the bucketed gradient all-reduce, against single-process computations on the full batch
(reference semantics: image_text_pretrain_loss.py:30-39,164-185; legacy_distributed_data_parallel.py:76-165)."""
import os
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, initfile, outdir):
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    try:
        from one_peace_amd.criterions.contrastive import contrastive_pair_loss, gather_without_grad
        from one_peace_amd.distributed import BucketedGradReducer, FlatParameters
        from oracle import onepeace_oracle as O
        torch.manual_seed(0)
        b, Hd = 3, 16
        full_a = torch.nn.functional.normalize(torch.randn(world * b, Hd), dim=1)
        full_t = torch.nn.functional.normalize(torch.randn(world * b, Hd), dim=1)
        a = full_a[rank * b:(rank + 1) * b].clone().requires_grad_(True)
        t = full_t[rank * b:(rank + 1) * b].clone().requires_grad_(True)
        t_all, a_all = gather_without_grad(t, a)
        assert torch.equal(t_all, full_t) and torch.equal(a_all, full_a), "rank-major gather order"
        assert not t_all.requires_grad
        scale = torch.tensor(10.0)
        loss, a_ok, t_ok = contrastive_pair_loss(a, t, a_all, t_all, scale, 0.1)
        ref, ra, rt = O.itc_loss(full_a[rank * b:(rank + 1) * b], full_t[rank * b:(rank + 1) * b], full_a, full_t, scale,
                                 rank=rank, label_smoothing=0.1)
        assert torch.allclose(loss, ref, atol=1e-6) and a_ok == ra and t_ok == rt
        loss.backward()
        assert a.grad is not None and t.grad is not None

        # bucketed all-reduce == mean of per-rank grads (LegacyDDP: divide by world, then SUM)
        torch.manual_seed(1)
        net = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
        ref_state = {k: v.clone() for k, v in net.state_dict().items()}
        flat = FlatParameters(net)
        red = BucketedGradReducer(flat, bucket_bytes=64)  # tiny buckets -> several collectives
        assert len(red.buckets) > 1
        x = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank))
        red.reset()
        net(x).pow(2).sum().backward()
        red.finish()
        mine = flat.grads.clone() / world
        torch.save({"avg": mine}, os.path.join(outdir, "r%d.pt" % rank))
        net2 = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
        net2.load_state_dict(ref_state)
        per_rank = []
        for r in range(world):
            net2.zero_grad()
            xr = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + r))
            net2(xr).pow(2).sum().backward()
            per_rank.append({n: p.grad.clone() for n, p in net2.named_parameters()})
        for (n, p, o, k) in flat.entries:
            want = O.dp_mean_grads([g[n] for g in per_rank])
            assert torch.allclose(mine[o:o + k].view(p.shape), want, atol=1e-6), n

        # --- a parameter shared by an autograd pass and a fused pass that accumulates in place (mixed use) ---
        from one_peace_amd import ops

        def run(late_after_autograd):
            net3 = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
            net3.load_state_dict(ref_state)
            fl = FlatParameters(net3)
            rd = BucketedGradReducer(fl, bucket_bytes=64)
            w = net3[1].weight
            delta = torch.full_like(w, 0.25 * (rank + 1))
            rd.reset()
            w._op_pending = 1                  # what ops._register_direct does in a fused forward
            xr = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank))
            net3(xr).pow(2).sum().backward()   # autograd contribution -> post-accumulate hook; w stays open (pending 1)
            b_of_w = rd._bucket_of[[n for n, *_ in fl.entries].index("1.weight")]
            assert not rd._launched[b_of_w], "bucket went out with an in-place contribution outstanding"
            if late_after_autograd:
                w.grad.add_(delta)             # the fused pass's in-place contribution arrives afterwards ...
                ops._direct_grad_done(w)       # ... and completes the parameter
                assert rd._launched[b_of_w]
            rd.finish()
            return fl, rd

        fl, rd = run(True)
        for (n, p_, o, k) in fl.entries:
            want = sum(g[n] for g in per_rank)
            if n == "1.weight":
                want = want + sum(torch.full_like(want, 0.25 * (r + 1)) for r in range(world))
            assert torch.allclose(fl.grads[o:o + k].view(p_.shape), want, atol=1e-5), n
        assert rd.stats["launched_in_backward"] == len(rd.buckets)
        dist.barrier()

        # --- a second backward pass without no_sync() must fail loudly, never re-reduce an already reduced bucket ---
        net6 = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
        fl6 = FlatParameters(net6)
        rd6 = BucketedGradReducer(fl6, bucket_bytes=64)
        rd6.reset()
        x6 = torch.randn(5, 8)
        net6(x6).pow(2).sum().backward()
        raised = False
        try:
            net6(x6).pow(2).sum().backward()
        except RuntimeError as e:
            raised = "second backward pass" in str(e)
        assert raised
        rd6.finish()
        dist.barrier()

        # --- gradient accumulation: micro-step 1 under no_sync(), micro-step 2 reduces the sum of both ---
        net4 = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
        net4.load_state_dict(ref_state)
        fl4 = FlatParameters(net4)
        rd4 = BucketedGradReducer(fl4, bucket_bytes=64)
        xa = torch.randn(5, 8, generator=torch.Generator().manual_seed(10 + rank))
        xb = torch.randn(5, 8, generator=torch.Generator().manual_seed(50 + rank))
        with rd4.no_sync():
            net4(xa).pow(2).sum().backward()
        rd4.reset()
        net4(xb).pow(2).sum().backward()
        rd4.finish()
        net5 = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.Linear(8, 4))
        net5.load_state_dict(ref_state)
        for r in range(world):
            for seed in (10 + r, 50 + r):
                net5(torch.randn(5, 8, generator=torch.Generator().manual_seed(seed))).pow(2).sum().backward()
        want = {n: q.grad for n, q in net5.named_parameters()}
        for (n, p_, o, k) in fl4.entries:
            assert torch.allclose(fl4.grads[o:o + k].view(p_.shape), want[n], atol=1e-5), n
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_world2_gather_loss_and_grad_allreduce():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        initfile = os.path.join(d, "init")
        mp.spawn(_worker, args=(world, initfile, d), nprocs=world, join=True)
        r0, r1 = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt"))
        assert torch.equal(r0["avg"], r1["avg"]), "ranks disagree after the all-reduce"


def test_world4_bench_control_flow_on_cpu():
    """bench.py's own data-parallel control flow -- env:// initialisation, parameter broadcast, three forwards, the fused [k,b,H]
    all-gather criterion, backward with the bucketed gradient all-reduce (46 buckets here) overlapped, clip + optimiser step,
    max-over-ranks timing and the JSON line -- at WORLD SIZE 4 over gloo (`--debug-cpu-micro`: 2-layer H=128 model, torch path,
    optim.TorchAdamW).  Every rank sees different data, so identical replicas after three steps mean every gradient was reduced
    after its last contribution; and because collectives pair up by issue order, every rank must have issued its bucket
    all-reduces in the same order (digest all-gathered by --check-replicas).  The multi-GPU tier is RCCL's first contact with
    more than one rank: this is what can be proven about it without the hardware."""
    import json
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1",
                        "--debug-cpu-micro", "--check-replicas"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "replica check: IDENTICAL" in r.stderr and "bucket launch order on every rank: IDENTICAL" in r.stderr, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints exactly one JSON line: %r" % lines
    out = json.loads(lines[0])
    d = out["config"]["distributed"]
    assert out["n_gpus"] == 4 and d["world_size_seen_by_backend"] == 4 and d["bucket_launch_order_identical_on_all_ranks"] is True
    assert out["config"]["global_batch"] == 4 * out["config"]["per_gpu_batch"] and out["scaling"] == "weak"
    ov = out["config"]["grad_allreduce_overlap"]
    assert ov["buckets"] == d["grad_buckets"] > 8 and ov["steps"] == 3
    # after the first step (which learns the parameters without a gradient) every bucket goes out DURING backward
    assert ov["launched_in_finish"] <= ov["buckets"] and ov["launched_in_backward"] >= 2 * ov["buckets"]
