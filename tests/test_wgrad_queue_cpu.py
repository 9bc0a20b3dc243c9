"""Host logic of the deferred weight-gradient queue (ops._WgradQueue), no GPU: launches are recorded by stand-ins for the two
ctypes wrappers.  Round-4 advisor findings: state left behind by a backward pass that raised; two contributions to one gradient
view inside one grouped launch; the kernel's alignment rule for C."""
import torch

from one_peace_amd import hip, ops


class _Rec:
    def __init__(self, monkeypatch):
        self.grouped, self.single = [], []
        monkeypatch.setattr(hip, "gemm_tn_grouped", lambda items, tune=0: self.grouped.append(list(items)) or True)
        monkeypatch.setattr(hip, "gemm_tn", lambda dy, x, out, acc: self.single.append((dy, x, out, acc)) or out)
        ops.reset_wgrads()


def _prob(rows=128, M=64, N=32, flat=None, off=0):
    dy, x = torch.zeros(rows, M, dtype=torch.bfloat16), torch.zeros(rows, N, dtype=torch.bfloat16)
    out = torch.zeros(M * N, dtype=torch.bfloat16) if flat is None else flat[off:off + M * N]
    return dy, x, out.view(M, N)


def test_queue_is_emptied_by_reset_and_rearms(monkeypatch):
    rec = _Rec(monkeypatch)
    q = ops._wgrad_queue
    p = torch.nn.Parameter(torch.zeros(1))
    w = torch.nn.Parameter(torch.zeros(4))

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            return a.clone()

        @staticmethod
        def backward(ctx, g):
            q.add(*_prob(), [p])           # a layer queued its problem ...
            raise RuntimeError("out of memory")  # ... and a later node of the same backward raised

    try:
        Boom.apply(w).sum().backward()
    except RuntimeError:
        pass
    assert q.items and q.armed, "the engine does not run queue_callback callbacks of a pass that raised"
    ops.reset_wgrads()
    assert not q.items and not q.done and not q.armed
    # the next pass arms the safety net again and flushes exactly its own problems at the end of backward
    class Ok(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            return a.clone()

        @staticmethod
        def backward(ctx, g):
            q.add(*_prob(), [p])
            q.add(*_prob(), [p])
            return g

    Ok.apply(w).sum().backward()
    assert len(rec.grouped) == 1 and len(rec.grouped[0]) == 2 and not q.items and not q.armed


def test_flat_zero_grad_drops_stale_problems(monkeypatch):
    from one_peace_amd.distributed import FlatParameters
    rec = _Rec(monkeypatch)
    lin = torch.nn.Linear(16, 8).to(torch.bfloat16)
    flat = FlatParameters(lin)
    ops._wgrad_queue.items.append(_prob() + (True,))
    ops._wgrad_queue.armed = True
    flat.zero_grad()
    assert not ops._wgrad_queue.items and not ops._wgrad_queue.armed and not rec.grouped


def test_second_contribution_to_a_view_goes_out_in_its_own_launch(monkeypatch):
    rec = _Rec(monkeypatch)
    q = ops._wgrad_queue
    flat = torch.zeros(3 * 64 * 32, dtype=torch.bfloat16)
    q.armed = True  # outside a backward pass no engine callback can be installed
    a, b, c = _prob(flat=flat, off=0), _prob(flat=flat, off=64 * 32), _prob(flat=flat, off=0)
    q.add(*a, [])
    q.add(*b, [])
    assert not rec.grouped
    q.add(*c, [])          # same view as `a`: unordered read-modify-write inside one launch would lose a contribution
    assert len(rec.grouped) == 1 and len(rec.grouped[0]) == 2 and len(q.items) == 1
    half = flat[32 * 32:32 * 32 + 64 * 32].view(64, 32)   # overlaps `a`'s second half only
    q.add(torch.zeros(128, 64, dtype=torch.bfloat16), torch.zeros(128, 32, dtype=torch.bfloat16), half, [])
    assert len(rec.grouped) == 1 and len(rec.single) == 1 and len(q.items) == 1  # the lone problem took the split-K launch
    q.flush()
    ops.reset_wgrads()


def test_wgrad_into_mirrors_the_kernels_alignment_rule(monkeypatch):
    rec = _Rec(monkeypatch)
    launched = []
    monkeypatch.setattr(ops, "wgrad", lambda dy, x, out=None, accumulate=False: launched.append(out))
    monkeypatch.setattr(hip, "gemm_tn_supported", lambda *a: True)
    flat = torch.zeros(64 * 40 + 16, dtype=torch.bfloat16)
    ops._wgrad_queue.armed = True
    dy, x = torch.zeros(128, 64, dtype=torch.bfloat16), torch.zeros(128, 40, dtype=torch.bfloat16)
    base = flat.data_ptr() % 16 // 2
    ok = flat[(8 - base) % 8:][:64 * 40].view(64, 40)
    ops.wgrad_into(dy, x, ok, [])
    assert len(ops._wgrad_queue.items) == 1 and not launched
    odd = flat[(8 - base) % 8 + 4:][:64 * 40].view(64, 40)       # 8-byte aligned only
    ops.wgrad_into(dy, x, odd, [])
    assert len(ops._wgrad_queue.items) == 1 and len(launched) == 1
    x36 = torch.zeros(128, 36, dtype=torch.bfloat16)
    ld36 = flat[(8 - base) % 8:][:64 * 36].view(64, 36)           # ldc % 8 == 4
    ops.wgrad_into(dy, x36, ld36, [])
    assert len(ops._wgrad_queue.items) == 1 and len(launched) == 2
    ops.reset_wgrads()


def _in_backward(fn):
    """Run fn() inside a backward pass (the queue arms an end-of-backward callback, which only exists there)."""
    class Run(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a):
            return a.clone()

        @staticmethod
        def backward(ctx, g):
            fn()
            return g
    Run.apply(torch.nn.Parameter(torch.zeros(2))).sum().backward()


def test_side_products_ride_on_the_grouped_launch_and_hooks_run_before_the_parameters_are_reported(monkeypatch):
    """Round 5 (layer-scale gradient from the weight gradient): a queued problem may carry (W, rowdot) and an `after` hook.  The side product
    travels with its problem into the grouped launch; the hook runs once the launch is enqueued and BEFORE the completion signals (gamma's
    own signal is given by the hook); a lone problem that misses the grouped launch takes the product once and uses it for the gradient
    and for the row dot; a reset after a failed pass drops the hooks (round 6: the side-product buffers are written once per launch, there is no pool to re-arm)."""
    rec = _Rec(monkeypatch)
    q = ops._wgrad_queue
    order = []
    p1, p2 = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
    for p in (p1, p2):
        p._op_pending = 1
        p._op_on_final = lambda param, tag=id(p): order.append(("final", tag))
    W, rowdot, gam = torch.ones(64, 32, dtype=torch.bfloat16), torch.full((2, 64), 9.0), torch.full((64,), 0.5, dtype=torch.bfloat16)

    def two():
        q.add(*_prob(), [p1], side=(W, rowdot, gam), after=lambda: order.append(("after", 0)))
        q.add(*_prob(), [p2])
    _in_backward(two)  # (flushed by the end-of-backward callback)
    assert len(rec.grouped) == 1 and rec.grouped[0][0][4] == (W, rowdot, gam) and rec.grouped[0][1][4] is None
    assert order[0] == ("after", 0) and [o[0] for o in order[1:]] == ["final", "final"]
    # a lone problem: no grouped launch; the product is formed once (accumulate = False into a temporary) and used twice
    rec.grouped.clear()
    made = []

    def fake_tn(dy, x, out, acc):
        made.append((out, acc))
        return torch.full((64, 32), 2.0, dtype=torch.bfloat16)
    monkeypatch.setattr(hip, "gemm_tn", fake_tn)
    dy, x, out = _prob()
    _in_backward(lambda: q.add(dy, x, out, [], side=(W, rowdot, gam), after=lambda: order.append(("after", 1))))
    assert not rec.grouped and made == [(None, False)]
    assert float(out.float().sum()) == 0.5 * 2.0 * 64 * 32 and torch.equal(rowdot.sum(0), torch.full((64,), 64.0))  # gradient gamma * P, row dot of the unscaled P
    assert order[-1] == ("after", 1)
    # reset after a failed pass: the hook of the abandoned problem never runs

    def failing():
        q.add(*_prob(), [], side=(W, rowdot, gam), after=lambda: order.append(("after", 2)))
        raise RuntimeError("out of memory")
    try:
        _in_backward(failing)
    except RuntimeError:
        pass
    assert q.items
    ops.reset_wgrads()
    q.flush()
    assert ("after", 2) not in order
