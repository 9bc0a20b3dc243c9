"""Autograd layer over the C-ABI HIP kernels (``hip.py``).

Each ``torch.autograd.Function`` here is what the host-side mirrors of the reference modules call when their
input is a bf16 tensor on an MI355X.  Forward and backward enqueue HIP kernels on the current stream; PyTorch
only owns the memory.  An encoder layer is two functions -- ``AttnBranchFn`` (LN, fused QKV GEMM, attention, sub-LN,
out-proj + layer-scale/drop-path residual) and ``FfnBranchFn`` (LN, GeGLU GEMM, LN(F), down GEMM + residual) -- so that
joint text+image / text+audio streams can route each modality's rows to its own FFN.  With ``save_acts=False`` a branch
keeps only its input and recomputes the intermediates in backward -- the reference's ``checkpoint_activations: true``
(pretrain_vl_3B.yaml:93, one_peace_pretrain.py:78-96); with ``save_acts=True`` (``checkpoint_activations: false``) it
keeps them, which 288 GB of HBM affords even for the 4B model at batch 64.
"""
import math
import os
import weakref

import torch

from . import hip

# --------------------------------------------------------------------------------------------------------------
# derived-buffer cache: transposed weights for the dgrad GEMMs (never parameters, rebuilt when the weight changes)
# --------------------------------------------------------------------------------------------------------------
_wt_cache = {}
_cache_epoch = 0


_refresh_plan = None  # (key set, device table, n, total tiles, [cache keys]) of the batched refresh


def invalidate_weight_cache():
    """Call after parameters were updated through raw pointers (the fused AdamW kernel does not bump _version)."""
    global _cache_epoch
    _cache_epoch += 1


def refresh_weight_cache():
    """After an optimiser step: rebuild EVERY cached transposed weight in one batched launch (520 matrices at 4B; lazily,
    one launch per matrix plus a concatenation per q/k/v triple, they cost 14 ms per step) and mark them current."""
    global _cache_epoch, _refresh_plan
    _cache_epoch += 1
    # fp8 copies of the FFN weights (opt-in variant): re-quantised IN PLACE into the same buffers -- a replayed TrainStepGraph keeps
    # reading these addresses, and an eager step must not pay a cache miss per weight either
    if not FP8_FFN and (_fp8_cache or _fp8_derived_cache):  # the opt-in variant was switched off: nothing reads these copies any more
        _fp8_cache.clear()
        _fp8_pairs.clear()
        _fp8_derived_cache.clear()
    for k, (ref, _, qs) in list(_fp8_cache.items()):
        w = ref()
        if w is None:
            del _fp8_cache[k]
            continue
        hip.quant_fp8_rows(w.detach(), out=qs)
        _fp8_cache[k] = (ref, (w._version, w.data_ptr(), _cache_epoch), qs)
    live = [(k, v) for k, v in _wt_cache.items() if all(r() is not None for r in v[0]) and (v[3] is None or v[3]() is not None)]
    if not live:
        return
    sig = tuple((k, tuple(r().data_ptr() for r in v[0]), v[2].data_ptr(), v[3]().data_ptr() if v[3] is not None else 0) for k, v in live)
    if _refresh_plan is None or _refresh_plan[0] != sig:
        jobs = []
        for k, v in live:
            off = 0
            for r in v[0]:
                w = r().detach()
                # segment i fills columns [off, off + out_i) of [in, sum out]; a scaled entry (one weight) carries its row scales
                jobs.append((w, v[2][:, off:off + w.shape[0]], v[3]().detach() if v[3] is not None else None))
                off += w.shape[0]
        table, tiles = hip.transpose_table(jobs, live[0][1][2].device)
        _refresh_plan = (sig, table, len(jobs), tiles)
    hip.transpose_batched(_refresh_plan[1], _refresh_plan[2], _refresh_plan[3])
    for k, v in live:
        _wt_cache[k] = (v[0], _wt_version(tuple(r() for r in v[0]), v[3]() if v[3] is not None else None), v[2], v[3])
    for k, (_, qs, tref) in list(_fp8_derived_cache.items()):  # fp8 copies of the transposed weights (opt-in fp8 input-gradient GEMMs)
        t = tref()
        if t is None:
            del _fp8_derived_cache[k]
            continue
        hip.quant_fp8_rows(t, out=qs)
        _fp8_derived_cache[k] = (_cache_epoch, qs, tref)


def _wt_version(ws, scale):
    return (tuple(w._version for w in ws), tuple(w.data_ptr() for w in ws), _cache_epoch,
            (scale._version, scale.data_ptr()) if scale is not None else None)


def _transposed(ws, scale=None):
    """[sum(out_i), in] -> cached bf16 [in, sum(out_i)] (ws: one weight or a tuple that is concatenated on dim 0).
    scale (bf16 [out], one weight only): column j of the copy is scale[j] * W[j, :] -- the last Linear of a residual branch with the
    layer scale folded in, the operand of  dx = (rowscale * dout) . (gamma o W)  when the branch gradient is kept un-scaled.

    Entries are keyed by the identity of the parameter objects and validated through weak references (a freed
    parameter's address can be handed to a different tensor by the allocator) plus (_version, epoch)."""
    ws = ws if isinstance(ws, (tuple, list)) else (ws,)
    assert scale is None or len(ws) == 1
    key = tuple(id(w) for w in ws) + ((id(scale),) if scale is not None else ())
    ver = _wt_version(ws, scale)
    hit = _wt_cache.get(key)
    if hit is not None and all(r() is w for r, w in zip(hit[0], ws)) and (scale is None or (hit[3] is not None and hit[3]() is scale)):
        if hit[1] == ver:
            return hit[2]
        out = hit[2]
    else:
        out = None
        if len(_wt_cache) > 4096:  # drop entries of parameters that no longer exist
            for k in [k for k, v in _wt_cache.items() if any(r() is None for r in v[0]) or (v[3] is not None and v[3]() is None)]:
                del _wt_cache[k]
    src = ws[0] if len(ws) == 1 else torch.cat([w.detach() for w in ws], dim=0)
    t = hip.transpose(src.detach(), out, scale=scale.detach() if scale is not None else None)
    _wt_cache[key] = (tuple(weakref.ref(w) for w in ws), ver, t, weakref.ref(scale) if scale is not None else None)
    return t


# --------------------------------------------------------------------------------------------------------------
# fp8 (e4m3) variant of the forward FFN GEMMs -- BASELINE configs[4], explicit opt-in (bench.py --config 4 --fp8)
# --------------------------------------------------------------------------------------------------------------
FP8_FFN = False
_fp8_cache = {}


FP8_FFN_DGRAD = os.environ.get("ONEPEACE_FP8_DGRAD", "1") != "0"


def set_fp8_ffn(on, dgrad=None):
    """Opt in / out of running the FFN GEMMs on fp8 e4m3 operands with per-row scales (csrc/fp8.hip): the forward up- and
    down-projection and (round 6; dgrad=False or ONEPEACE_FP8_DGRAD=0 keeps them in bf16) the two INPUT-gradient GEMMs of the backward
    -- d LN_F(g) = dy W2 and d LN2(x) = dh [W0 | W1] -- whose operands are quantised row by row (the gradient rows by op_quant_fp8_rows,
    the transposed weight copies once per optimiser step).  The WEIGHT gradients stay on the bf16 kernels and the bf16 activations."""
    global FP8_FFN, FP8_FFN_DGRAD
    old, FP8_FFN = FP8_FFN, bool(on)
    if dgrad is not None:
        FP8_FFN_DGRAD = bool(dgrad)
    return old


_fp8_derived_cache = {}


def _fp8_derived(t):
    """(fp8 bytes, row scales) of a DERIVED bf16 matrix that keeps its address across optimiser steps (a cached transposed weight copy,
    _transposed).  Quantised on first use; refresh_weight_cache re-quantises every entry IN PLACE right behind the transposes it rebuilds
    (eagerly, outside any captured graph: a replayed TrainStepGraph keeps reading these addresses and must find this step's weights);
    after invalidate_weight_cache (lazy mode) the next use re-quantises."""
    key = (t.data_ptr(), tuple(t.shape))
    hit = _fp8_derived_cache.get(key)
    if hit is not None and hit[0] == _cache_epoch and hit[2]() is t:
        return hit[1]
    if len(_fp8_derived_cache) > 4096:
        _fp8_derived_cache.clear()
        hit = None
    qs = hip.quant_fp8_rows(t, out=hit[1] if hit is not None else None)
    _fp8_derived_cache[key] = (_cache_epoch, qs, weakref.ref(t))
    return qs


# --------------------------------------------------------------------------------------------------------------
# memory level between "keep every activation" and the reference's checkpoint_activations
# --------------------------------------------------------------------------------------------------------------
RECOMPUTE_CHEAP = os.environ.get("ONEPEACE_RECOMPUTE_CHEAP", "0") == "1"


def set_recompute_cheap(on):
    """With save_acts=True, do NOT keep the four row matrices a LayerNorm-type pass can re-create from tensors that are kept anyway:
    LN1(x), the attention sub-LayerNorm's output, LN2(x_mid) and LN_F(gelu(h0) * h1) -- they are read only by the weight-gradient
    GEMMs.  Backward re-runs the same kernels on the same inputs (bit-identical operands, so bit-identical gradients): 14 H of the
    46 H bytes kept per token and layer = 63 GB at the headline batch (128 tuples x 571 tokens x 40 layers) for 3 LayerNorm + 1
    LN-GeGLU forward passes per layer in backward (+33 ms of a 700 ms step).  bench.py switches it on instead of halving the per-GPU
    batch when RCCL's buffers leave too little room on a multi-GPU node (pretrain_vl_3B.yaml:93 checkpoint_activations is the
    reference's only level)."""
    global RECOMPUTE_CHEAP
    old, RECOMPUTE_CHEAP = RECOMPUTE_CHEAP, bool(on)
    return old


def _fp8_weight(w):
    """(fp8 bytes, row scales) of a weight [out, in]; a derived buffer like the transposed dgrad copies, re-quantised IN PLACE when
    the weight changes (parameter version / optimiser epoch): the buffers keep their addresses."""
    key = id(w)
    ver = (w._version, w.data_ptr(), _cache_epoch)
    hit = _fp8_cache.get(key)
    if hit is not None and hit[0]() is w:
        if hit[1] != ver:
            hip.quant_fp8_rows(w.detach(), out=hit[2])
            _fp8_cache[key] = (hit[0], ver, hit[2])
        return hit[2]
    if len(_fp8_cache) > 4096:
        for k in [k for k, v in _fp8_cache.items() if v[0]() is None]:
            del _fp8_cache[k]
    qs = hip.quant_fp8_rows(w.detach())
    _fp8_cache[key] = (weakref.ref(w), ver, qs)
    return qs


_fp8_pairs = {}


def _fp8_weight_pair(w0, w1):
    """wi_0 | wi_1 as ONE fp8 matrix [2F, H] + scales [2F] (the plain N = 2F up-projection of the training forward): the two weights'
    cache entries are views of the halves, so the per-weight refresh keeps the pair current in place."""
    key = (id(w0), id(w1))
    hit = _fp8_pairs.get(key)
    if hit is None or hit[0]() is not w0 or hit[1]() is not w1:
        F0, K = w0.shape
        q = torch.empty(F0 + w1.shape[0], K, dtype=torch.uint8, device=w0.device)
        sc = torch.empty(F0 + w1.shape[0], dtype=torch.float32, device=w0.device)
        for w, lo, hi in ((w0, 0, F0), (w1, F0, q.shape[0])):
            view = (q[lo:hi], sc[lo:hi])
            hip.quant_fp8_rows(w.detach(), out=view)
            _fp8_cache[id(w)] = (weakref.ref(w), (w._version, w.data_ptr(), _cache_epoch), view)
        if len(_fp8_pairs) > 1024:
            for k in [k for k, v in _fp8_pairs.items() if v[0]() is None or v[1]() is None]:
                del _fp8_pairs[k]
        hit = _fp8_pairs[key] = (weakref.ref(w0), weakref.ref(w1), (q, sc))
    else:
        _fp8_weight(w0)
        _fp8_weight(w1)
    return hit[2]


def _round_up(n, m):
    return ((n + m - 1) // m) * m


def _t_pad(x2d):
    """[M, C] -> [C, Mpad] (Mpad = M rounded up to 64, zero tail): the K-contiguous operand of a weight-gradient GEMM."""
    M, C = x2d.shape
    Mp = _round_up(M, 64)
    out = torch.empty(C, Mp, dtype=x2d.dtype, device=x2d.device)
    if Mp != M:
        out[:, M:].zero_()
    hip.transpose(x2d, out)
    return out


def _wgrad(dyT, xT):
    """dW[out, in] = dy^T x from the two transposed, K-padded operands."""
    return hip.gemm_nt(dyT, [xT])


USE_TN_WGRAD = True


_tail_pads = {}


def _tail_pad(t, rows0, operand):
    """The last len(t) - rows0 (< 64) rows of `t` as a 64-row matrix whose other rows are zero.  The buffer is cached per
    (device, stream, operand, row count, width): it is zeroed once and only its first rows are ever overwritten."""
    r, cols = t.shape[0] - rows0, t.shape[1]
    key = (t.device, torch.cuda.current_stream(t.device).cuda_stream, operand, r, cols)
    buf = _tail_pads.get(key)
    if buf is None:
        if len(_tail_pads) > 64:
            _tail_pads.clear()
        buf = _tail_pads[key] = torch.zeros(64, cols, dtype=t.dtype, device=t.device)
    buf[:r].copy_(t[rows0:])
    return buf


def wgrad_tn_rows(K):
    """Rows of a weight-gradient GEMM the transpose-read kernel takes in its main launch (it consumes 64 rows per step)."""
    return K - K % 64


def wgrad(dy, x, out=None, accumulate=False):
    """dW[out, in] (+)= dy[rows, out]^T x[rows, in].  Transpose-read GEMM straight from the row-major activations; a row count
    that is not a multiple of 64 (16 x 257 image tokens) is split into the multiple-of-64 part and a second, one-step launch
    over a zero-padded copy of the < 64 leftover rows, accumulated into the same output.  Shapes the kernel does not take
    at all (fewer than 64 rows, widths that are not multiples of 8): two transposed K-padded copies + the NT kernel."""
    K, M = dy.shape
    N = x.shape[1]
    K0 = wgrad_tn_rows(K)
    if USE_TN_WGRAD and K0 >= 64 and hip.gemm_tn_supported(K0, M, N, dy.stride(0), x.stride(0)):
        out = hip.gemm_tn(dy[:K0], x[:K0], out, accumulate)
        if K0 != K:
            hip.gemm_tn(_tail_pad(dy, K0, 0), _tail_pad(x, K0, 1), out, True)
        return out
    if out is not None and accumulate:
        return out.add_(hip.gemm_nt(_t_pad(dy), [_t_pad(x)]))
    return hip.gemm_nt(_t_pad(dy), [_t_pad(x)], out=out)


# --------------------------------------------------------------------------------------------------------------
# deferred weight gradients: all dW GEMMs of an encoder layer as ONE grouped launch (csrc/gemm.hip: gemm256w_tn_grouped_kernel)
# --------------------------------------------------------------------------------------------------------------
GROUPED_WGRAD = os.environ.get("ONEPEACE_GROUPED_WGRAD", "1") != "0"


class _WgradQueue:
    """Weight-gradient GEMMs that accumulate into flat gradient views, collected while a layer's backward runs (FFN branch
    first, then the attention branch, which flushes): launched alone each of them has 36 ... 288 output tiles for 256 CUs and
    needs split-K slabs + a fold; together they are 1440 tiles for one persistent launch without split-K.  The operands are
    kept alive until the launch is enqueued; the reducer hears about a parameter only after that."""

    def __init__(self):
        self.items, self.done, self.after, self.armed, self.epoch = [], [], [], False, 0

    def reset(self):
        """Forget everything a backward pass that raised left behind (the engine does not run queue_callback callbacks then):
        stale operands must not be accumulated into the freshly zeroed gradients of the next step, and the end-of-backward
        safety net has to be registered again.  Called by distributed.FlatParameters.zero_grad / BucketedGradReducer.reset."""
        self.items, self.done, self.after, self.armed = [], [], [], False
        self.epoch += 1

    @staticmethod
    def _span(out):
        """Byte range [lo, hi) an output view covers in its storage (rows may be strided)."""
        lo = out.data_ptr()
        return lo, lo + ((out.shape[0] - 1) * out.stride(0) + out.shape[1]) * out.element_size()

    def add(self, dy, x, out, params, side=None, after=None):
        """side: (W, rowdot [N / 128, M], gamma) -- `dy` carries NO layer scale: the launch adds gamma[m] * product[m] to out[m] and writes the
        partial sums of W[m][n] * product[m][n] over 128-column slots to rowdot (hip.gemm_tn_grouped: rscale);
        after: called once the launch that holds this problem is enqueued (before the parameters are reported)."""
        lo, hi = self._span(out)
        for it in self.items:  # the grouped launch read-modify-writes C tiles without ordering between problems: two
            l2, h2 = self._span(it[2])  # contributions to one (overlapping) view go out as two launches, stream-ordered
            if lo < h2 and l2 < hi:
                self.flush()
                break
        self.items.append((dy, x, out, True, side))
        self.done.extend(params)
        if after is not None:
            self.after.append(after)
        if not self.armed:  # safety net: whatever is still queued when autograd finishes this backward pass goes out then
            self.armed = True
            epoch = self.epoch
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self._end_of_backward(epoch))
        if len(self.items) >= hip.TN_GROUP_MAX:
            self.flush()

    def _end_of_backward(self, epoch):
        if epoch != self.epoch:  # a reset() came in between: this callback belongs to an abandoned pass
            return
        self.armed = False
        self.flush()

    def flush(self):
        items, done, after = self.items, self.done, self.after
        self.items, self.done, self.after = [], [], []
        if items:
            if len(items) == 1 or not hip.gemm_tn_grouped(items):  # a lone problem keeps the split-K launch of op_gemm_tn
                for dy, x, out, _, side in items:
                    if side is None:
                        hip.gemm_tn(dy, x, out, True)
                    else:  # (rare: the side product rides on the grouped launch only) the product once, used twice
                        prod = hip.gemm_tn(dy, x, None, False)
                        W, rowdot, gamma = side
                        out.addcmul_(prod, gamma.detach().unsqueeze(1))
                        rowdot.zero_()  # [slots, M]: the whole row dot goes into slot 0
                        for r0 in range(0, prod.shape[0], 256):  # (row blocks: no fp32 copies of whole matrices)
                            rowdot[0, r0:r0 + 256] = (W.detach()[r0:r0 + 256].float() * prod[r0:r0 + 256].float()).sum(1)
        for f in after:
            f()
        for q in done:
            _direct_grad_done(q)


_wgrad_queue = _WgradQueue()


def _wgrad_queueable(K, M, N, ldy, ldx, grad_view):
    return (GROUPED_WGRAD and USE_TN_WGRAD and K % 64 == 0 and K >= 64 and grad_view.stride(0) % 8 == 0 and grad_view.stride(1) == 1
            and grad_view.data_ptr() % 16 == 0  # op_gemm_tn_grouped's rule for C: ldc % 8 == 0, 16-byte aligned (else launch now)
            and hip.gemm_tn_supported(K, M, N, ldy, ldx))


def wgrad_into(dy, x, grad_view, params, side=None, after=None):
    """grad_view (+)= dy^T x for the flat-buffer gradient views of `params` (one view spanning all of them).  Deferred into the
    layer's grouped launch when the shape allows, else launched now; either way every parameter's completion is signalled.
    side / after: see _WgradQueue.add (the caller has checked dgamma_from_wgrad_ok: the problem IS queueable)."""
    K, M = dy.shape
    if _wgrad_queueable(K, M, x.shape[1], dy.stride(0), x.stride(0), grad_view):
        _wgrad_queue.add(dy, x, grad_view, params, side, after)
        return
    assert side is None and after is None, "wgrad_into: a side product was promised for a problem the grouped launch does not take"
    wgrad(dy, x, out=grad_view, accumulate=True)
    for q in params:
        _direct_grad_done(q)


# --------------------------------------------------------------------------------------------------------------
# layer-scale gradient without the branch output (round 5)
# --------------------------------------------------------------------------------------------------------------
# out = resid + ps * gamma * y with y = x W^T + b (transformer_layer.py:70-88).  dgamma[n] = sum_m ps dout[m][n] y[m][n] needs y in
# backward -- a second output of the residual GEMM (its epilogue then moves three times the bytes of a plain launch) that is kept for
# backward (2 H bytes per token and branch) and read again by op_resid_bwd.  But
#     dgamma[n] = sum_k W[n][k] * G[n][k] + b[n] * g0[n],     G = (ps dout)^T x,   g0 = sum_m ps dout,
# and the weight gradient the step computes anyway is dW = gamma[n] * G.  (Round 6) The branch gradient op_resid_bwd writes is
# u = ps dout WITHOUT gamma; the grouped weight-gradient launch multiplies u^T x = G (its fp32 accumulators) with W into a row vector
# -- that IS the first term, for any gamma including 0 -- and adds gamma[n] * G to the bf16 gradient; the input gradient
# u . (gamma o W) reads a gamma-scaled transposed copy of W (rebuilt with the others after every optimiser step).  Round 5 fed the
# launch the gamma-scaled gradient and divided the row vector by gamma: wrong (0) where gamma == 0.  y is neither written nor kept.
DGAMMA_FROM_WGRAD = os.environ.get("ONEPEACE_DGAMMA_FROM_WGRAD", "1") != "0"


def _rowdot_slots(device, weights):
    """The side-product buffer of a branch: fp32 [sum_i in_i / 128, out] -- weight i ([out, in_i]: the last Linear of one weight set
    that shares gamma) owns in_i / 128 consecutive slots, each written exactly once by the grouped launch (no zeroing, no atomics).
    Returns (buffer, [per-weight [in_i / 128, out] views])."""
    n = [w.shape[1] // 128 for w in weights]
    buf = torch.empty(sum(n), weights[0].shape[0], dtype=torch.float32, device=device)
    views, off = [], 0
    for k in n:
        views.append(buf[off:off + k])
        off += k
    return buf, views


def dgamma_from_wgrad_ok(rows, gamma, weights, biases, needs_g, needs_w):  # rows: per weight, the token rows of its gradient GEMM
    """Forward-time decision (it fixes whether the branch output is written at all): gamma and every last-Linear weight / bias of the
    branch accumulate in place in the flat gradient buffer, all of them want a gradient, and the weight gradients will ride on the
    grouped launch with full 256 x 256 tiles."""
    if not (DGAMMA_FROM_WGRAD and needs_g and all(needs_w) and gamma is not None and _direct_grad(gamma) and gamma.dtype == torch.bfloat16):
        return False
    for k, w in zip(rows, weights):
        if not (_direct_grad(w) and w.is_contiguous() and w.shape[0] % 256 == 0 and w.shape[1] % 256 == 0 and w.data_ptr() % 16 == 0
                and _wgrad_queueable(k, w.shape[0], w.shape[1], w.shape[0], w.shape[1], w.grad)):
            return False
    return all(b is None or _direct_grad(b) for b in biases)


def flush_wgrads():
    _wgrad_queue.flush()


def reset_wgrads():
    """Drop weight-gradient problems a failed backward pass left queued (see _WgradQueue.reset)."""
    _wgrad_queue.reset()


def _direct_grad(param):
    """True when `param` lives in distributed.FlatParameters: its .grad is a pre-allocated, pre-zeroed view the weight
    gradient GEMM can accumulate into directly (no autograd accumulation pass, no temporary dW tensor)."""
    return getattr(param, "_op_flat", False) and param.grad is not None


def _direct_grad_done(param):
    """One backward contribution to a directly accumulated gradient finished; the last one notifies the reducer."""
    left = getattr(param, "_op_pending", 1) - 1
    param._op_pending = left
    if left <= 0:
        cb = getattr(param, "_op_on_final", None)
        if cb is not None:
            cb(param)


def gemm_any(A, W, bias=None, out_f32=False, alpha=None):
    """A[M,K] @ W[N,K]^T (+bias) through the HIP GEMM for ANY K, N (zero-pads K to 64 / N to 8 on the host)."""
    M, K = A.shape
    N = W.shape[0]
    Kp, Np = _round_up(K, 64), _round_up(N, 8)
    if Kp != K:
        A = torch.nn.functional.pad(A, (0, Kp - K))
        W = torch.nn.functional.pad(W, (0, Kp - K))
    if Np != N:
        W = torch.nn.functional.pad(W, (0, 0, 0, Np - N))
        if bias is not None:
            bias = torch.nn.functional.pad(bias, (0, Np - N))
    A = A.contiguous()
    W = W.contiguous()
    if out_f32:
        out = hip.gemm_nt(A, [W], [bias] if bias is not None else None, epilogue=hip.EPI_F32, alpha=alpha)
    else:
        out = hip.gemm_nt(A, [W], [bias] if bias is not None else None)
    return out if Np == N else out[:, :N].contiguous()


def hip_eligible(x):
    """The HIP path takes bf16 CUDA tensors; anything else runs the torch reference ops of the mirrors."""
    return x.is_cuda and x.dtype == torch.bfloat16


# --------------------------------------------------------------------------------------------------------------
# LayerNorm (+ optional fused GELU)
# --------------------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, gelu):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        need = any(ctx.needs_input_grad[:3])
        y, mean, rstd = hip.layernorm_fwd(x2, w, b, eps, gelu=gelu, want_stats=need)
        if need:
            ctx.save_for_backward(x2, w, b, mean, rstd)
            ctx.gelu = gelu
        return y.view(shape)

    @staticmethod
    def backward(ctx, dy):
        x2, w, b, mean, rstd = ctx.saved_tensors
        dy2 = dy.reshape(x2.shape)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx, dw, db = hip.layernorm_bwd(dy2, x2, w, b, mean, rstd, gelu=ctx.gelu,
                                       need_wgrad=w is not None and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]))
        return dx.view(dy.shape), dw, db, None, None


def layer_norm(x, w, b, eps=1e-5, gelu=False):
    return LayerNormFn.apply(x, w, b, eps, gelu)


# --------------------------------------------------------------------------------------------------------------
# Linear  y = x W^T + b
# --------------------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y = gemm_any(x2, w, b)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        return y.view(*shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.reshape(-1, w.shape[0])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_any(dy2, _transposed(w)).view(*dy.shape[:-1], w.shape[1])
        if ctx.needs_input_grad[1]:
            if dy2.shape[1] % 8 == 0 and x2.shape[1] % 8 == 0:
                dw = wgrad(dy2, x2)
            else:
                dw = gemm_any(_t_pad(dy2), _t_pad(x2))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            dy8 = dy2 if dy2.shape[1] % 8 == 0 else torch.nn.functional.pad(dy2, (0, 8 - dy2.shape[1] % 8))
            db = hip.colsum(dy8.contiguous())[: w.shape[0]]
        return dx, dw, db


def linear(x, w, b=None):
    return LinearFn.apply(x, w, b)


# --------------------------------------------------------------------------------------------------------------
# relative-position bias image  [heads][S][Spad]  (and its transpose for the dK/dV kernel)
# --------------------------------------------------------------------------------------------------------------
class RelPosBias:
    """Per-forward handle: bf16 images of table[bucket] plus the fp32 accumulator the attention backward adds into."""

    def __init__(self, table, bucket_i32, S, ids=None):
        """ids: optional int32 [B, K] position ids (masked pretraining: a different kept-token subset per sample) -- the images
        are then per sample, [B, heads, K, Kpad], built straight from the table (no dense [B, heads, S, S] tensor, no gathers),
        and S is K."""
        self.ids = ids
        self.S, self.Spad = S, hip.attn_spad(S)
        self.num_rel, self.heads = table.shape
        self.bucket = bucket_i32
        self.table = table
        self.acc = None
        self._imageT = None
        self._frag = None
        self.image = _RelPosImageFn.apply(table, self)

    @property
    def frag(self):
        """Fragment-major copy of the image for the resident forward kernel (built on first use, shared by all layers)."""
        if self._frag is None:
            self._frag = hip.attn_bias_pack(self.image.detach(), self.S)
        return self._frag

    @property
    def imageT(self):
        """out[h][key][query]: what the dK/dV kernel reads (built on first use in a backward pass)."""
        if self._imageT is None:
            if self.ids is not None:
                self._imageT = hip.relpos_bias_build_ids(self.table.detach(), self.bucket, self.ids, self.Spad, transposed=True)
            else:
                self._imageT = hip.relpos_bias_build(self.table.detach(), self.bucket, self.S, self.Spad, transposed=True)
        return self._imageT

    def grad_accumulator(self, B):
        """fp32 [slabs, heads, S, Spad]: the attention backward of every layer that uses this table adds its dS sums here
        (per-sample images: one slab per sample)."""
        if self.acc is None:
            self.acc = hip.attn_dbias_buffer(B, self.S, self.heads, self.Spad, self.image.device, per_sample=self.ids is not None)
        elif self.ids is None:  # layers that skip their dropped samples call with different batch sizes: slab counts differ
            need = hip.lib().op_attn_bwd_dbias_slabs(B, self.S, self.heads, hip.TUNE.attn_bwd())
            if need > self.acc.shape[0]:
                grown = torch.zeros(need, *self.acc.shape[1:], dtype=self.acc.dtype, device=self.acc.device)
                grown[:self.acc.shape[0]] = self.acc
                self.acc = grown
        return self.acc


class DenseBias:
    """Handle for a PER-SAMPLE additive bias [B, heads, S, S] (what the adapters' gather of a different token subset per
    sample produces in masked pretraining, adapter/image.py:229-246 / common.take_bias): same interface as RelPosBias, but
    the images are [B, heads, S, Spad], and the attention backward returns one gradient slab per sample, which flows back
    into the dense tensor (and through autograd's gather into the table)."""

    def __init__(self, dense):
        B, self.heads, self.S, _ = dense.shape
        self.B, self.Spad = B, hip.attn_spad(self.S)
        self.acc = None
        self._imageT = None
        self._frag = None
        self.image = _DenseBiasFn.apply(dense, self)

    @property
    def frag(self):
        if self._frag is None:
            self._frag = hip.attn_bias_pack(self.image.detach(), self.S)
        return self._frag

    @property
    def imageT(self):
        if self._imageT is None:
            img = self.image.detach()
            t = torch.zeros_like(img)
            t[..., : self.S] = img[..., : self.S].transpose(2, 3)
            self._imageT = t
        return self._imageT

    def grad_accumulator(self, B):
        if self.acc is None:
            self.acc = hip.attn_dbias_buffer(B, self.S, self.heads, self.Spad, self.image.device, per_sample=True)
        return self.acc


class _DenseBiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dense, handle):
        ctx.handle_ref = weakref.ref(handle)
        ctx.S = dense.shape[-1]
        B, heads, S, _ = dense.shape
        img = torch.zeros(B, heads, S, hip.attn_spad(S), dtype=dense.dtype, device=dense.device)
        img[..., :S] = dense
        return img

    @staticmethod
    def backward(ctx, _unused):
        h = ctx.handle_ref()
        if h is None or h.acc is None:
            return torch.zeros(_unused.shape[:-1] + (ctx.S,), dtype=_unused.dtype, device=_unused.device), None
        g = h.acc[..., : ctx.S].to(_unused.dtype)
        h.acc = None
        return g, None


class _RelPosImageFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, handle):
        ctx.handle_ref = weakref.ref(handle)  # the layers' ctx keep the handle alive until their backward ran
        ctx.shape = tuple(table.shape)
        if handle.ids is not None:
            return hip.relpos_bias_build_ids(table, handle.bucket, handle.ids, handle.Spad)
        return hip.relpos_bias_build(table, handle.bucket, handle.S, handle.Spad)

    @staticmethod
    def backward(ctx, _unused):
        # The attention backward kernels add dS straight into handle.acc (fp32); the tensor gradient that autograd
        # routes here is a placeholder that only orders this node after every consuming layer.
        h = ctx.handle_ref()
        if h is None or h.acc is None:
            return torch.zeros(ctx.shape, dtype=torch.bfloat16, device=_unused.device), None
        if h.ids is not None:
            dtable = hip.relpos_bias_bwd_ids(h.acc, h.bucket, h.ids, h.num_rel)
        else:
            dtable = hip.relpos_bias_bwd(h.acc.sum(0), h.bucket, h.num_rel, h.S, h.Spad)
        h.acc = None
        return dtable.to(torch.bfloat16), None


# --------------------------------------------------------------------------------------------------------------
# the fused encoder layer
# --------------------------------------------------------------------------------------------------------------
_DIRECT_WEIGHTS = ("wq", "wk", "wv", "wo", "w0", "w1", "w2")
ATTN_PARAMS = ("ln1_w", "ln1_b", "wq", "bq", "wk", "wv", "bv", "aln_w", "aln_b", "wo", "bo", "g1")
FFN_PARAMS = ("ln2_w", "ln2_b", "w0", "w1", "fln_w", "fln_b", "w2", "b2", "g2")
LAYER_PARAMS = ATTN_PARAMS + FFN_PARAMS


class StreamSeg:
    """One stream's rows inside a packed [sum B*S, H] activation matrix: B samples of S tokens starting at row `row0`, with their
    own relative-position bias handle (RelPosBias-like or None) and key-padding bytes.  A single-stream layer is one segment;
    a lock-step pass over several modalities (transformer_encoder.forward_multi) has one per modality.  pad: rows behind the B*S
    token rows that still belong to the segment (zero rows of a packed kept-sample matrix, hip.KeptRows): LayerNorms and GEMMs run
    over them, the attention core does not."""
    __slots__ = ("name", "B", "S", "row0", "bias", "key_pad", "pad")

    def __init__(self, name, B, S, row0, bias=None, key_pad=None, pad=0):
        self.name, self.B, self.S, self.row0, self.bias, self.key_pad, self.pad = name, B, S, row0, bias, key_pad, pad

    @property
    def tokens(self):
        return self.B * self.S

    @property
    def rows(self):
        return self.B * self.S + self.pad

    @property
    def end(self):
        return self.row0 + self.B * self.S + self.pad

    def frag(self, limit):
        return self.bias.frag if self.bias is not None and self.S <= limit else None


def _attn_forward(x2, P, segs, heads, scale, rowscale, rps, keep, want_y=True, xmap=None, out=None):
    """x_mid = x + rowscale * g1 * out_proj(subLN(attention(LN1(x))))  on the packed rows x2 [N, H]: LayerNorm, the fused q|k|v
    projection, the sub-LayerNorm and the output projection run over ALL rows in one launch each, the attention core per
    segment.  rowscale: fp32 drop-path multipliers indexed by row // rps (None = 1).  want_y: also write the branch output y1
    (only the gradient of gamma_1 needs it).  xmap (hip.KeptRows.rowmap) + out: x2 is the FULL matrix, the branch runs on its rows
    xmap[r] and writes x_mid for them into `out` (a full matrix; the other rows are the caller's)."""
    H = x2.shape[1]
    xln1, mean1, rstd1 = hip.layernorm_fwd(x2, P["ln1_w"], P["ln1_b"], want_stats=keep, x_rows=xmap)
    if H % 128 == 0:
        Nr = xln1.shape[0]
        full = _qkv_rows_that_fill_whole_rounds(Nr, 3 * H) if QKV_ROUND_SPLIT else Nr
        if full < Nr:  # (689.8 / 689.7 -> 687.3 / 687.0 ms on the headline step, same box: profiles/r5_experiments.md section 12)
            qkv = torch.empty(Nr, 3 * H, dtype=x2.dtype, device=x2.device)
            for lo, hi in ((0, full), (full, Nr)):
                hip.gemm_nt(xln1[lo:hi], [P["wq"], P["wk"], P["wv"]], [P["bq"], None, P["bv"]], n_seg=H, N=3 * H, out=qkv[lo:hi])
        else:
            qkv = hip.gemm_nt(xln1, [P["wq"], P["wk"], P["wv"]], [P["bq"], None, P["bv"]], n_seg=H, N=3 * H)
    else:
        qkv = torch.empty(xln1.shape[0], 3 * H, dtype=x2.dtype, device=x2.device)
        for i, (w, b) in enumerate(((P["wq"], P["bq"]), (P["wk"], None), (P["wv"], P["bv"]))):
            hip.gemm_nt(xln1, [w], [b] if b is not None else None, out=qkv[:, i * H:(i + 1) * H], ldc=3 * H)
    attn = torch.empty_like(xln1)
    lses = []
    for sg in segs:
        r = slice(sg.row0, sg.end)
        _, lse = hip.attn_fwd(qkv[r, :H], qkv[r, H:2 * H], qkv[r, 2 * H:], 3 * H, sg.B, sg.S, heads, scale,
                              sg.bias.image.detach() if sg.bias is not None else None, sg.key_pad, hip.attn_spad(sg.S), out=attn[r],
                              want_lse=keep, bias_frag=sg.frag(hip.ATTN_RESIDENT_MAX_S))
        lses.append(lse)
        if sg.pad:  # rows of no sample: the out-proj weight gradient multiplies them with exact zeros, so they must be finite
            attn[sg.end - sg.pad:sg.end].zero_()
    if P["aln_w"] is not None:
        aln, mean_a, rstd_a = hip.layernorm_fwd(attn, P["aln_w"], P["aln_b"], want_stats=keep)
    else:
        aln, mean_a, rstd_a = attn, None, None
    y1 = torch.empty_like(xln1) if keep and want_y else None
    x_mid = hip.gemm_nt(aln, [P["wo"]], [P["bo"]], epilogue=hip.EPI_RESID, resid=x2, gamma=P["g1"], rowscale=rowscale,
                        rows_per_sample=rps, h0=y1, out=out, resid_rows=xmap)
    if not keep:
        return x_mid, None
    acts = dict(xln1=xln1, mean1=mean1, rstd1=rstd1, qkv=qkv, attn=attn, aln=aln, mean_a=mean_a, rstd_a=rstd_a, y1=y1)
    for i, lse in enumerate(lses):
        acts["lse%d" % i] = lse
    return x_mid, acts


QKV_ROUND_SPLIT = os.environ.get("ONEPEACE_QKV_ROUND_SPLIT", "1") != "0"
# (round 6, ABI 9) a branch that runs on the samples stochastic depth keeps (hip.KeptRows) reads and writes the FULL activation /
# gradient matrix through the row table (KeptRows.rowmap: LayerNorm forward / backward, op_resid_bwd, the residual epilogue of the
# branch's last GEMM) instead of through packed copies -- no op_rows_gather / op_rows_merge passes (17.4 ms of the 586 ms step with
# skip_dropped_branches, profiles/r6_bench_skip_dropped_last_step.txt), the same bits.  "0": round 4's packed copies (A/B, tests).
SKIP_ROW_TABLES = os.environ.get("ONEPEACE_SKIP_ROW_TABLES", "1") != "0"


def _row_tables(kept):
    return kept is not None and SKIP_ROW_TABLES and not FP8_FFN


def _qkv_rows_that_fill_whole_rounds(rows, n_out, cus=256, max_tail=512):
    """Largest multiple of 256 rows whose 256 x 256 tiles are a whole number of rounds of `cus` workgroups, when that leaves at most
    `max_tail` rows for a second, small launch and saves a round (73 088 rows x 4608 columns: 5 148 tiles = 20.1 rounds -- 28 tiles
    cost a 21st round; 284 row tiles = 5 112 tiles = 20 rounds + a 384-row launch); else `rows`."""
    tn, tm = (n_out + 255) // 256, (rows + 255) // 256
    rounds = -(-tm * tn // cus)
    tm_main = (rounds - 1) * cus // tn
    full = tm_main * 256
    return full if 0 < rows - full <= max_tail and tm_main > 0 else rows


GEGLU_SPLIT = os.environ.get("ONEPEACE_GEGLU_SPLIT", "1") != "0"
SEPARATE_DELTA = os.environ.get("ONEPEACE_SEPARATE_DELTA", "0") == "1"  # A/B switch: op_attn_bwd_delta pass instead of the fused one


def _geglu_split(Fd, fln_w):
    """Training forward of the GeGLU: plain two-segment GEMM writing h0 | h1 + op_ln_geglu_fwd instead of the EPI_GEGLU epilogue
    (which keeps the erf on the GEMM's critical path: one wave per SIMD cannot hide ~16 VALU per output element) followed by a
    LayerNorm pass over g.  Needs the inner sub-LayerNorm (scale_fc) and whole 256-column tiles per weight segment."""
    return GEGLU_SPLIT and fln_w is not None and Fd % 256 == 0


def _ffn_forward(x_mid, P, S, ps2, keep, want_y=True, grad=None):
    """out = x_mid + ps2 * g2 * W2(LN_F(gelu(LN2(x_mid) W0^T) * (LN2(x_mid) W1^T)))  on x_mid [B*S, H].
    grad: the pass belongs to a training step (default: keep).  The GeGLU form is chosen by THAT, not by `keep`: with
    checkpoint_activations the forward (keep = False) and its recomputation in backward (keep = True) must run the same kernels, or the
    LayerNorm(F) statistics the gradients use are not those of the forward output.  No-grad inference keeps the fused
    GeGLU epilogue (one store instead of three; h0 / h1 rounded to bf16 only in the split form: a bf16-level train / eval difference)."""
    if grad is None:
        grad = keep
    Fd = P["w0"].shape[0]
    h0 = h1 = None
    fp8 = FP8_FFN and x_mid.shape[1] % 128 == 0 and Fd % 128 == 0
    split = grad and _geglu_split(Fd, P["fln_w"])
    xq8 = gq8 = None
    if fp8 and split:  # (round 5) the fp8 operands come out of the kernels that produce the bf16 rows: no quantisation passes
        xln2, mean2, rstd2, xq8 = hip.layernorm_fwd(x_mid, P["ln2_w"], P["ln2_b"], want_stats=keep, q8=True)
    else:
        xln2, mean2, rstd2 = hip.layernorm_fwd(x_mid, P["ln2_w"], P["ln2_b"], want_stats=keep)
    if keep and not split:
        h0 = torch.empty(x_mid.shape[0], Fd, dtype=x_mid.dtype, device=x_mid.device)
        h1 = torch.empty_like(h0)
    if split:  # plain wi_0 | wi_1 up-projection; GELU, gate and the inner LayerNorm in one HBM-bound pass
        if fp8:  # (round 5: the fp8 variant takes the same route -- its GELU no longer sits in a GEMM epilogue either)
            w01q, w01s = _fp8_weight_pair(P["w0"], P["w1"])
            hh = hip.gemm_nt_fp8(xq8[0], xq8[1], [w01q], [w01s])
            gq8 = (torch.empty(hh.shape[0], Fd, dtype=torch.uint8, device=hh.device), torch.empty(hh.shape[0], dtype=torch.float32, device=hh.device))
        else:
            hh = hip.gemm_nt(xln2, [P["w0"], P["w1"]], n_seg=Fd, N=2 * Fd)
        h0, h1 = hh[:, :Fd], hh[:, Fd:]
        gln, mean_f, rstd_f = hip.ln_geglu_fwd(h0, h1, P["fln_w"], P["fln_b"], q8=gq8)
    elif fp8:  # opt-in: e4m3 operands with per-row scales, fp32 accumulation (csrc/fp8.hip)
        xq, xs = hip.quant_fp8_rows(xln2)
        (w0q, w0s), (w1q, w1s) = _fp8_weight(P["w0"]), _fp8_weight(P["w1"])
        g = hip.gemm_nt_fp8(xq, xs, [w0q, w1q], [w0s, w1s], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)
    else:
        g = hip.gemm_nt(xln2, [P["w0"], P["w1"]], epilogue=hip.EPI_GEGLU, h0=h0, h1=h1)
    if split:
        pass
    elif P["fln_w"] is not None:
        gln, mean_f, rstd_f = hip.layernorm_fwd(g, P["fln_w"], P["fln_b"], want_stats=keep)
    else:
        gln, mean_f, rstd_f = g, None, None
    y2 = torch.empty_like(x_mid) if keep and want_y else None
    if fp8:
        gq, gs = gq8 if gq8 is not None else hip.quant_fp8_rows(gln)
        w2q, w2s = _fp8_weight(P["w2"])
        out = hip.gemm_nt_fp8(gq, gs, [w2q], [w2s], bias=P["b2"], epilogue=hip.EPI_RESID, resid=x_mid, gamma=P["g2"], rowscale=ps2,
                              rows_per_sample=S, h0=y2)
    else:
        out = hip.gemm_nt(gln, [P["w2"]], [P["b2"]], epilogue=hip.EPI_RESID, resid=x_mid, gamma=P["g2"], rowscale=ps2,
                          rows_per_sample=S, h0=y2)
    if not keep:
        return out, None
    # g (the GeGLU output) is not kept: the fused LN(F)+GeGLU backward recomputes it from h0, h1
    return out, dict(xln2=xln2, mean2=mean2, rstd2=rstd2, h0=h0, h1=h1, gln=gln, mean_f=mean_f, rstd_f=rstd_f, y2=y2)


def _register_direct(ctx, names, params, needs):
    """Weights whose gradient GEMM accumulates straight into the flat gradient buffer (distributed.FlatParameters).
    Keeps (name, Parameter object) pairs: saved_tensors hands back fresh tensor objects without .grad / attributes."""
    ctx.direct = tuple((n, q) for n, q, ng in zip(names, params, needs) if ng and q is not None and _direct_grad(q))
    for _, q in ctx.direct:
        q._op_pending = getattr(q, "_op_pending", 0) + 1


def _targets(direct, *names):
    """(grad views, accumulate) for a kernel that produces the gradients of `names` together: in-place accumulation
    only when every one of them lives in the flat gradient buffer."""
    if all(n in direct for n in names):
        return [direct[n].grad for n in names], True
    return [None] * len(names), False


def _finish(direct, G, names, values, accumulated):
    for n, v in zip(names, values):
        if accumulated:
            _direct_grad_done(direct[n])
        else:
            G[n] = v


def _return_grads(names, params, G, direct):
    """Gradients handed back to autograd: None for absent parameters and for the ones accumulated in place -- unless a
    producer fell back to a temporary, which is then folded into the flat buffer here."""
    out = []
    for n, q in zip(names, params):
        g = G.get(n) if q is not None else None
        if g is not None and n in direct:
            direct[n].grad.add_(g.view_as(direct[n].grad))
            _direct_grad_done(direct[n])
            g = None
        out.append(g)
    return out


def _resid_backward(dout, y, gamma, ps, S, gname, bname, direct, G, needs, g0=None, rows=None):
    """Gradient of  resid + ps * gamma * (y)  w.r.t. the branch output and -- where they are wanted -- gamma and the last
    Linear's bias, in one pass.  g0 (fp32 [H]): gamma's gradient is NOT taken here (no y: dgamma_from_wgrad_ok); the kernel fills g0
    with sum_m ps * dout for op_gamma_grad_finish instead and returns ps * dout WITHOUT gamma."""
    names = tuple(n for n, present in ((gname, gamma is not None and g0 is None), (bname, True)) if present and needs.get(n))
    tgt, acc = _targets(direct, *names) if names else ([], False)
    t = dict(zip(names, tgt))
    dgamma = (t[gname] if acc else True) if gname in t else None
    dbias = (t[bname] if acc else True) if bname in t else None
    dy, dg, db = hip.resid_bwd(dout, y if dgamma is not None else None, gamma, ps, S, dgamma=dgamma, dbias=dbias, accumulate=acc, g0=g0,
                               dout_rows=rows)  # (rows: dout is the full matrix, the pass runs on its rows rows[m])
    _finish(direct, G, names, tuple({gname: dg, bname: db}[n] for n in names), acc)
    return dy


def _adjacent_grads(direct, names):
    """The parameters `names` in the memory order of their flat gradient views when those views tile ONE contiguous range
    (distributed.FlatParameters lays a layer's decayed weights out back to back), else None.  A merged weight-gradient GEMM
    then writes all of them with one launch (and one split-K fold) instead of one launch each."""
    if not all(n in direct for n in names):
        return None
    order = sorted(names, key=lambda n: direct[n].grad.data_ptr())
    for a, b in zip(order, order[1:]):
        ga, gb = direct[a].grad, direct[b].grad
        if not ga.is_contiguous() or ga.data_ptr() + ga.numel() * ga.element_size() != gb.data_ptr() or ga.shape[1:] != gb.shape[1:]:
            return None
    return order


def _span(direct, order):
    """One [sum rows, cols] view over the adjacent gradient views of `order`."""
    g0 = direct[order[0]].grad
    return torch.as_strided(g0, (sum(direct[n].grad.shape[0] for n in order), g0.shape[1]), (g0.shape[1], 1))


def _weight_grad_fn(ctx, G):
    direct = dict(ctx.direct)

    def weight_grad(name, dyv, xv, side=None, after=None):
        target = direct.get(name)
        if target is not None:
            wgrad_into(dyv, xv, target.grad, (target,), side, after)
        else:
            assert side is None and after is None
            G[name] = wgrad(dyv, xv)
    return weight_grad, direct


def _gamma_finish_hook(rowdot, gamma_param, pairs, done):
    """After the grouped launch: dgamma += sum_s rowdot[s] + sum b_i g0_i into gamma's flat gradient view, and gamma's in-place
    contribution is reported (the reducer may all-reduce its bucket)."""
    def run():
        hip.gamma_grad_finish(rowdot, pairs, gamma_param.grad, True)
        if done:
            _direct_grad_done(gamma_param)
    return run


def _save(ctx, keep, acts, *tensors):
    if keep:  # 288 GB of HBM: keep the intermediates instead of recomputing them in backward
        ctx.act_names = [k for k, v in acts.items() if v is not None]
        ctx.save_for_backward(*tensors, *[acts[k] for k in ctx.act_names])
    else:
        ctx.act_names = None
        ctx.save_for_backward(*tensors)


def _restore(ctx, saved_acts, optional):
    A = dict(zip(ctx.act_names, saved_acts))
    for k in optional:
        A.setdefault(k, None)
    return A


class AttnBranchFn(torch.autograd.Function):
    """First half of transformer_layer.py:165-228: x + droppath(gamma_1 * self_attn(LN(x))), forward + backward in HIP, over the
    packed rows of one or more streams.

    x2: [N, H] bf16 contiguous rows (batch-major per segment; a segment may itself be a joint text+image / text+audio
    stream).  segs: [StreamSeg] covering the N rows.  rowscale: fp32 drop-path multipliers (0 or 1/keep) indexed by
    row // rps, or None.  imgs: one bias image per segment (or None) -- only there to put the tables into the autograd graph."""

    @staticmethod
    def forward(ctx, x2, segs, rowscale, rps, heads, save_acts, kept, *rest):
        nseg = len(segs)
        params = rest[nseg:]
        P = dict(zip(ATTN_PARAMS, params))
        need_grad = any(ctx.needs_input_grad) and bool(int(save_acts) & 2)  # bit 1: autograd is recording (_save_flags) -- a
        keep = bool(int(save_acts) & 1) and need_grad                       # no-grad teacher pass registers and keeps nothing
        first_param = 7 + nseg
        needs = dict(zip(ATTN_PARAMS, ctx.needs_input_grad[first_param:]))
        x_full = x2
        mapped = _row_tables(kept)  # the kept samples' rows through the row table (no packed copy of x, no merge pass)
        if kept is not None and not mapped:  # the branch runs on the rows of the samples it keeps (segs / rowscale describe THOSE rows)
            x2 = hip.rows_gather(x_full, kept)
        H = x2.shape[1]
        N = kept.total if mapped else x2.shape[0]
        scale = (H // heads) ** -0.5
        # gamma_1's gradient from the out-proj weight gradient instead of from the branch output (then y1 is never written)
        ctx.dg_fused = need_grad and dgamma_from_wgrad_ok([N], P["g1"], [P["wo"]], [P["bo"]], needs["g1"], [needs["wo"]])
        if mapped:
            x_mid = torch.empty_like(x_full)
            _, acts = _attn_forward(x_full, P, segs, heads, scale, rowscale, rps, keep, want_y=needs["g1"] and not ctx.dg_fused,
                                    xmap=kept.rowmap(), out=x_mid)
            hip.rows_merge(x_full, None, kept, out=x_mid)  # (the dropped samples' rows: copied)
        else:
            x_mid, acts = _attn_forward(x2, P, segs, heads, scale, rowscale, rps, keep, want_y=needs["g1"] and not ctx.dg_fused)
            if kept is not None:
                x_mid = hip.rows_merge(x_full, x_mid, kept)
        ctx.kept, ctx.mapped = kept, mapped
        ctx.segs, ctx.dims, ctx.n_params, ctx.first_param = segs, (N, H, heads, scale, rps), len(params), first_param
        ctx.direct = ()
        if need_grad:
            _register_direct(ctx, ATTN_PARAMS, params, ctx.needs_input_grad[first_param:])
        if keep:  # activations only a frozen parameter's gradient would read are not kept
            if not needs["wo"]:
                acts["aln"] = None if P["aln_w"] is not None else acts["aln"]
            if not (needs["wq"] or needs["wk"] or needs["wv"]):
                acts["xln1"] = None
        ctx.cheap = keep and bool(int(save_acts) & 4)
        if ctx.cheap:  # (set_recompute_cheap) re-created in backward from x2 / attn, which are kept anyway
            acts["xln1"] = None
            if P["aln_w"] is not None:
                acts["aln"] = None
        _save(ctx, keep, acts, x2, rowscale, *params)
        return x_mid

    @staticmethod
    def backward(ctx, dx_mid):
        x2, rowscale, *rest = ctx.saved_tensors
        params, saved_acts = rest[:ctx.n_params], rest[ctx.n_params:]
        N, H, heads, scale, rps = ctx.dims
        segs = ctx.segs
        nseg = len(segs)
        P = dict(zip(ATTN_PARAMS, params))
        # what autograd actually asks for (frozen parameters -- stage-2 audio-language pretraining freezes the whole attention
        # branch, one_peace_pretrain.py:98-104 -- cost no weight-gradient GEMM, column sum or LayerNorm parameter reduction)
        needs = {n: bool(ng) and q is not None for n, q, ng in zip(ATTN_PARAMS, params, ctx.needs_input_grad[ctx.first_param:])}
        need_x = bool(ctx.needs_input_grad[0])
        want_dbias = [sg.bias is not None and sg.bias.image.requires_grad for sg in segs]
        kept, mapped = ctx.kept, ctx.mapped
        xmap = kept.rowmap() if mapped else None  # (mapped: x2 is the FULL matrix the forward read through the table)
        if ctx.act_names is not None:
            A = _restore(ctx, saved_acts, ("mean_a", "rstd_a", "y1", "xln1", "aln"))
            if ctx.cheap:  # the weight-gradient operands that were not kept: the same kernels on the same rows, bit for bit
                if needs["wq"] or needs["wk"] or needs["wv"]:
                    A["xln1"] = hip.layernorm_fwd(x2, P["ln1_w"], P["ln1_b"], x_rows=xmap)[0]
                if needs["wo"] and P["aln_w"] is not None:
                    A["aln"] = hip.layernorm_fwd(A["attn"], P["aln_w"], P["aln_b"])[0]
        else:  # recompute (the reference's checkpoint_activations behaviour)
            _, A = _attn_forward(x2, P, segs, heads, scale, rowscale, rps, True, want_y=not ctx.dg_fused, xmap=xmap,
                                 out=torch.empty_like(x2) if mapped else None)
        if not dx_mid.is_contiguous():
            dx_mid = dx_mid.contiguous()
        dx_full = dx_mid
        if kept is not None and not mapped:  # rows of dropped samples: the gradient passes through the skip connection untouched
            dx_mid = hip.rows_gather(dx_full, kept)
        G = {}
        weight_grad, direct = _weight_grad_fn(ctx, G)
        if ctx.dg_fused:  # dy1 = rowscale * dx_mid, WITHOUT gamma_1 (weight gradient: rscale; input gradient: scaled weight copy)
            g0 = torch.empty(H, dtype=torch.float32, device=dx_mid.device)
            dy1 = _resid_backward(dx_mid, None, P["g1"], rowscale, rps, "g1", "bo", direct, G, needs, g0=g0, rows=xmap)
            rowdot, (rd_wo,) = _rowdot_slots(dx_mid.device, [P["wo"]])
            weight_grad("wo", dy1, A["aln"], side=(P["wo"], rd_wo, P["g1"]),
                        after=_gamma_finish_hook(rowdot, direct["g1"], [(P["bo"], g0)] if P["bo"] is not None else [], True))
            wo_t = _transposed(P["wo"], scale=P["g1"])
        else:
            dy1 = _resid_backward(dx_mid, A["y1"], P["g1"], rowscale, rps, "g1", "bo", direct, G, needs, rows=xmap)
            if needs["wo"]:
                weight_grad("wo", dy1, A["aln"])
            wo_t = None
        upstream = ("ln1_w", "ln1_b", "wq", "bq", "wk", "wv", "bv", "aln_w", "aln_b")
        dx = None
        if need_x or any(want_dbias) or any(needs[n] for n in upstream):
            daln = hip.gemm_nt(dy1, [wo_t if wo_t is not None else _transposed(P["wo"])])
            if P["aln_w"] is not None:
                want = needs["aln_w"] or needs["aln_b"]
                (tw, tb), acc = _targets(direct, "aln_w", "aln_b") if want else ((None, None), False)
                dattn, dw_, db_ = hip.layernorm_bwd(daln, A["attn"], P["aln_w"], P["aln_b"], A["mean_a"], A["rstd_a"], dw=tw, db=tb,
                                                    accumulate=acc, need_wgrad=want)
                if want:
                    _finish(direct, G, ("aln_w", "aln_b"), (dw_, db_), acc)
            else:
                dattn = daln
            # column order of the packed q|k|v gradient: the memory order of the three weights' flat gradient views when those
            # are adjacent (then ONE weight-gradient launch writes all three), q, k, v otherwise
            qkv_names = ("wq", "wk", "wv")
            order = _adjacent_grads(direct, qkv_names) if all(needs[n] for n in qkv_names) and H % 8 == 0 else None
            cols = tuple(order) if order else qkv_names
            slot = {n: i for i, n in enumerate(cols)}
            qkv = A["qkv"]
            dqkv = torch.empty_like(qkv)
            for i, sg in enumerate(segs):
                r = slice(sg.row0, sg.end)
                if sg.pad:  # rows of no sample: zero gradient (they are operand rows of the q|k|v weight gradient)
                    dqkv[sg.end - sg.pad:sg.end].zero_()
                dparts = {n: dqkv[r, slot[n] * H:(slot[n] + 1) * H] for n in qkv_names}
                _attn_backward(qkv[r], dattn[r], A["attn"][r], A["lse%d" % i], sg.B, sg.S, heads, scale,
                               sg.bias.image.detach() if sg.bias is not None else None, sg.bias.imageT if sg.bias is not None else None,
                               sg.key_pad, sg.bias.grad_accumulator(sg.B) if want_dbias[i] else None, sg.frag(384),
                               dparts["wq"], dparts["wk"], dparts["wv"])
            bias_of = {"wq": "bq", "wv": "bv"}
            bnames = tuple(bias_of[n] for n in cols if n in bias_of and needs[bias_of[n]])
            if bnames:
                tgt, acc = _targets(direct, *bnames)
                if H % 8 == 0:
                    outs = [None, None, None]
                    for n, t_ in zip(bnames, tgt):
                        outs[slot["wq" if n == "bq" else "wv"]] = t_ if acc else torch.empty(H, dtype=dqkv.dtype, device=dqkv.device)
                    hip.colsum_segments(dqkv, H, outs, accumulate=acc)
                    _finish(direct, G, bnames, tuple(outs[slot["wq" if n == "bq" else "wv"]] for n in bnames), acc)
                else:
                    sums = hip.colsum(dqkv)
                    for n in bnames:
                        i = slot["wq" if n == "bq" else "wv"]
                        G[n] = sums[i * H:(i + 1) * H]
            if order:  # one problem: dW[3H, H] straight into the three adjacent flat gradient views
                wgrad_into(dqkv, A["xln1"], _span(direct, order), [direct[n] for n in order])
            elif any(n in direct for n in qkv_names) or not all(needs[n] for n in qkv_names):
                for n in qkv_names:
                    if needs[n]:
                        weight_grad(n, dqkv[:, slot[n] * H:(slot[n] + 1) * H], A["xln1"])
            else:
                dW = wgrad(dqkv, A["xln1"])  # [3H, H]
                G["wq"], G["wk"], G["wv"] = dW[:H], dW[H:2 * H], dW[2 * H:]
            if need_x or needs["ln1_w"] or needs["ln1_b"]:
                dxln1 = hip.gemm_nt(dqkv, [_transposed(tuple(P[n] for n in cols))])
                want = needs["ln1_w"] or needs["ln1_b"]
                (tw, tb), acc = _targets(direct, "ln1_w", "ln1_b") if want else ((None, None), False)
                # (mapped: the kept samples' rows of a NEW full matrix -- the incoming gradient may be the caller's tensor --, the dropped
                # samples' rows copied below)
                dx, dw_, db_ = hip.layernorm_bwd(dxln1, x2, P["ln1_w"], P["ln1_b"], A["mean1"], A["rstd1"], add=dx_mid, dw=tw, db=tb,
                                                 accumulate=acc, need_wgrad=want, dx=torch.empty_like(dx_full) if mapped else None,
                                                 x_rows=xmap)
                if want:
                    _finish(direct, G, ("ln1_w", "ln1_b"), (dw_, db_), acc)
        if dx is None and need_x:  # nothing upstream of the residual wanted a gradient: only the skip connection carries one
            dx = dx_full
        elif dx is not None and mapped:
            hip.rows_merge(dx_full, None, kept, out=dx)
        elif dx is not None and kept is not None:
            dx = hip.rows_merge(dx_full, dx, kept)
        flush_wgrads()  # the layer's weight gradients (this branch's and the FFN branch's, which ran before it) as one launch
        grads = _return_grads(ATTN_PARAMS, params, G, direct)
        dimgs = []
        for sg, w in zip(segs, want_dbias):  # placeholders (see _RelPosImageFn.backward); the real gradients went into bias.acc
            # ONE placeholder per handle and backward pass: the engine counts the image node's dependencies by graph edges (an undefined
            # gradient still decrements them), so the node runs after every consuming layer either way -- forty placeholders made autograd
            # materialise and add forty [heads, S, Spad] zero images per stream and step (3 fills + 3 adds per layer in the step trace)
            if w and not getattr(sg.bias, "_grad_routed", False):
                sg.bias._grad_routed = True
                img = sg.bias.image
                dimgs.append(torch.zeros((), dtype=img.dtype, device=img.device).expand(img.shape))
            else:
                dimgs.append(None)
        return (dx, None, None, None, None, None, None, *dimgs, *grads)


class FfnBranchFn(torch.autograd.Function):
    """Second half of transformer_layer.py:165-228: x + droppath(gamma_2 * modality_ffn(LN(x))) for the rows of ONE
    modality.  x: [B, S, H] contiguous; ps: fp32 [B] or None."""

    @staticmethod
    def forward(ctx, x, ps, save_acts, *params):
        B, S, H = x.shape
        P = dict(zip(FFN_PARAMS, params))
        x2 = x.reshape(B * S, H)
        need_grad = any(ctx.needs_input_grad) and bool(int(save_acts) & 2)
        keep = bool(int(save_acts) & 1) and need_grad
        needs = dict(zip(FFN_PARAMS, ctx.needs_input_grad[3:]))
        # (needs_input_grad is also set inside torch.no_grad() -- the teacher passes of a trainable model: only the flag says "recording")
        ctx.dg_fused = need_grad and dgamma_from_wgrad_ok([B * S], P["g2"], [P["w2"]], [P["b2"]], needs["g2"], [needs["w2"]])
        out, acts = _ffn_forward(x2, P, S, ps, keep, want_y=needs["g2"] and not ctx.dg_fused, grad=bool(int(save_acts) & 2))
        ctx.dims, ctx.n_params = (B, S, H), len(params)
        ctx.direct = ()
        if need_grad:
            _register_direct(ctx, FFN_PARAMS, params, ctx.needs_input_grad[3:])
        if keep:
            if not needs["w2"] and P["fln_w"] is not None:
                acts["gln"] = None
            if not (needs["w0"] or needs["w1"]):
                acts["xln2"] = None
        ctx.cheap = keep and bool(int(save_acts) & 4)
        if ctx.cheap:  # (set_recompute_cheap; also with the fp8 forward: backward reads the bf16 rows, which the same kernels re-create)
            acts["xln2"] = None
            if _geglu_split(P["w0"].shape[0], P["fln_w"]):  # LN_F(gelu(h0) * h1) comes back from the kept h0 | h1
                acts["gln"] = None
        _save(ctx, keep, acts, x2, ps, *params)
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, dout):
        x_mid, ps, *rest = ctx.saved_tensors
        params, saved_acts = rest[:ctx.n_params], rest[ctx.n_params:]
        B, S, H = ctx.dims
        P = dict(zip(FFN_PARAMS, params))
        needs = {n: bool(ng) and q is not None for n, q, ng in zip(FFN_PARAMS, params, ctx.needs_input_grad[3:])}
        need_x = bool(ctx.needs_input_grad[0])
        if ctx.act_names is not None:
            A = _restore(ctx, saved_acts, ("mean_f", "rstd_f", "y2", "gln", "xln2"))
            if ctx.cheap:
                if needs["w0"] or needs["w1"]:
                    A["xln2"] = hip.layernorm_fwd(x_mid, P["ln2_w"], P["ln2_b"])[0]
                if needs["w2"] and A["gln"] is None:
                    A["gln"] = hip.ln_geglu_fwd(A["h0"], A["h1"], P["fln_w"], P["fln_b"])[0]
        else:
            _, A = _ffn_forward(x_mid, P, S, ps, True, want_y=not ctx.dg_fused)
        N = B * S
        Fd = P["w0"].shape[0]
        dout2 = dout.reshape(N, H)
        if not dout2.is_contiguous():
            dout2 = dout2.contiguous()
        G = {}
        weight_grad, direct = _weight_grad_fn(ctx, G)
        if ctx.dg_fused:  # dy2 = ps * dout, WITHOUT gamma_2 (see AttnBranchFn.backward)
            g0 = torch.empty(H, dtype=torch.float32, device=dout2.device)
            dy2 = _resid_backward(dout2, None, P["g2"], ps, S, "g2", "b2", direct, G, needs, g0=g0)
            rowdot, (rd_w2,) = _rowdot_slots(dout2.device, [P["w2"]])
            weight_grad("w2", dy2, A["gln"], side=(P["w2"], rd_w2, P["g2"]),
                        after=_gamma_finish_hook(rowdot, direct["g2"], [(P["b2"], g0)] if P["b2"] is not None else [], True))
            w2_t = _transposed(P["w2"], scale=P["g2"])
        else:
            dy2 = _resid_backward(dout2, A["y2"], P["g2"], ps, S, "g2", "b2", direct, G, needs)
            if needs["w2"]:
                weight_grad("w2", dy2, A["gln"])
            w2_t = None
        dx = None
        if need_x or any(needs[n] for n in ("ln2_w", "ln2_b", "w0", "w1", "fln_w", "fln_b")):
            fp8b = FP8_FFN and FP8_FFN_DGRAD and H % 128 == 0 and Fd % 128 == 0  # (opt-in) the two input-gradient GEMMs on e4m3 operands
            w2t = w2_t if w2_t is not None else _transposed(P["w2"])
            if fp8b:
                dgln = hip.gemm_nt_fp8(*hip.quant_fp8_rows(dy2), *[[x] for x in _fp8_derived(w2t)])
            else:
                dgln = hip.gemm_nt(dy2, [w2t])
            # dh0 | dh1 as the two halves of ONE [N, 2F] matrix, in the memory order of the two weights' flat gradient views:
            # one weight-gradient launch (288 output tiles, one fold) when those views are adjacent, and always one K = 2F
            # input-gradient GEMM instead of two K = F launches chained through a residual epilogue
            order = _adjacent_grads(direct, ("w0", "w1")) if needs["w0"] and needs["w1"] and Fd % 8 == 0 else None
            cols = tuple(order) if order else ("w0", "w1")
            dh = torch.empty(N, 2 * Fd, dtype=dgln.dtype, device=dgln.device)
            dpart = {n: dh[:, i * Fd:(i + 1) * Fd] for i, n in enumerate(cols)}
            if P["fln_w"] is not None:  # sub-LayerNorm(F) and GeGLU backward in one pass; g is recomputed from h0, h1
                want = needs["fln_w"] or needs["fln_b"]
                (tw, tb), acc = _targets(direct, "fln_w", "fln_b") if want else ((None, None), False)
                _, _, dw_, db_ = hip.ln_geglu_bwd(dgln, A["h0"], A["h1"], P["fln_w"], A["mean_f"], A["rstd_f"], dw=tw, db=tb,
                                                  accumulate=acc, need_wgrad=want, dh0=dpart["w0"], dh1=dpart["w1"])
                if want:
                    _finish(direct, G, ("fln_w", "fln_b"), (dw_, db_), acc)
            else:
                d0, d1 = hip.geglu_bwd(dgln, A["h0"], A["h1"])
                dpart["w0"].copy_(d0)
                dpart["w1"].copy_(d1)
            if order:
                wgrad_into(dh, A["xln2"], _span(direct, order), [direct[n] for n in order])
            else:
                for n in ("w0", "w1"):
                    if needs[n]:
                        weight_grad(n, dpart[n], A["xln2"])
            if need_x or needs["ln2_w"] or needs["ln2_b"]:
                w01t = _transposed(tuple(P[n] for n in cols))
                if fp8b:
                    dxln2 = hip.gemm_nt_fp8(*hip.quant_fp8_rows(dh), *[[x] for x in _fp8_derived(w01t)])
                else:
                    dxln2 = hip.gemm_nt(dh, [w01t])
                want = needs["ln2_w"] or needs["ln2_b"]
                (tw, tb), acc = _targets(direct, "ln2_w", "ln2_b") if want else ((None, None), False)
                dx, dw_, db_ = hip.layernorm_bwd(dxln2, x_mid, P["ln2_w"], P["ln2_b"], A["mean2"], A["rstd2"], add=dout2, dw=tw, db=tb,
                                                 accumulate=acc, need_wgrad=want)
                if want:
                    _finish(direct, G, ("ln2_w", "ln2_b"), (dw_, db_), acc)
        if dx is None and need_x:
            dx = dout2
        grads = _return_grads(FFN_PARAMS, params, G, direct)
        return (dx.view(B, S, H) if dx is not None else None, None, None, *grads)


FFN_SHARED = ("ln2_w", "ln2_b", "g2")                       # shared by every modality (final_layer_norm, gamma_2)
FFN_OWN = ("w0", "w1", "fln_w", "fln_b", "w2", "b2")          # one set per modality (text_ffn / image_ffn / audio_ffn)


class FfnBranchMultiFn(torch.autograd.Function):
    """Second half of transformer_layer.py:165-228 for a lock-step pass: x + droppath(gamma_2 * <modality>_ffn(LN(x))) on the packed
    rows of several modalities, each segment through its OWN FFN weights.  final_layer_norm (forward and backward) runs over all
    rows in one launch; the two narrow GEMMs of the branch -- down-projection + residual (N = H, K = F) and the input gradient of
    wi_0 | wi_1 (N = H, K = 2F) -- are ONE grouped launch over the modalities (op_gemm_nt_grouped: a text pass alone is 192 tiles
    on 256 CUs, the image pass 3.02 rounds); the wide ones (GeGLU up-projection, its LN(F), the input gradient of the
    down-projection) and the weight gradients stay per modality.

    x2: [N, H]; segs: [StreamSeg] (only row0 / B / S are used); pss: per segment fp32 [B] drop-path multipliers or None;
    params: ln2_w, ln2_b, g2, then (w0, w1, fln_w, fln_b, w2, b2) per segment."""

    @staticmethod
    def _compute(x2, segs, pss, params, keep, want_y, grad=None, xmap=None):
        """xmap (hip.KeptRows.rowmap, not with the fp8 FFN): x2 is the FULL matrix, the branch runs on its rows xmap[r]; the returned
        `out` is a full matrix in which the rows of the kept samples are written (the others are the caller's)."""
        nseg = len(segs)
        shared = dict(zip(FFN_SHARED, params[:3]))
        own = [dict(zip(FFN_OWN, params[3 + 6 * i:9 + 6 * i])) for i in range(nseg)]
        H = x2.shape[1]
        N = xmap.numel() if xmap is not None else x2.shape[0]
        Fd = own[0]["w0"].shape[0]
        dev, dt = x2.device, x2.dtype
        has_fln = own[0]["fln_w"] is not None
        split = (keep if grad is None else grad) and _geglu_split(Fd, own[0]["fln_w"])  # (by the kind of pass, not by `keep`: see _ffn_forward)
        fp8 = FP8_FFN and split and H % 256 == 0 and Fd % 256 == 0  # (round 5) the opt-in fp8 forward: training (split) form only
        if fp8:  # all rows of all streams: the fp8 operand comes out of the LayerNorm kernel itself
            xln2, mean2, rstd2, (xq, xs) = hip.layernorm_fwd(x2, shared["ln2_w"], shared["ln2_b"], want_stats=keep, q8=True)
            gq, gs = torch.empty(N, Fd, dtype=torch.uint8, device=dev), torch.empty(N, dtype=torch.float32, device=dev)
        else:
            xln2, mean2, rstd2 = hip.layernorm_fwd(x2, shared["ln2_w"], shared["ln2_b"], want_stats=keep, x_rows=xmap)
        if split:
            hh = torch.empty(N, 2 * Fd, dtype=dt, device=dev)
            g, h0, h1 = None, hh[:, :Fd], hh[:, Fd:]
        else:
            g = torch.empty(N, Fd, dtype=dt, device=dev)
            h0 = torch.empty(N, Fd, dtype=dt, device=dev) if keep else None
            h1 = torch.empty(N, Fd, dtype=dt, device=dev) if keep else None
        gln = torch.empty(N, Fd, dtype=dt, device=dev) if has_fln else g
        mean_f = torch.empty(N, dtype=torch.float32, device=dev) if (keep or split) and has_fln else None
        rstd_f = torch.empty(N, dtype=torch.float32, device=dev) if (keep or split) and has_fln else None
        L = hip.lib()
        for sg, P in zip(segs, own):
            r = slice(sg.row0, sg.end)
            if split:
                if fp8:
                    w01q, w01s = _fp8_weight_pair(P["w0"], P["w1"])
                    hip.gemm_nt_fp8(xq[r], xs[r], [w01q], [w01s], out=hh[r])
                else:
                    hip.gemm_nt(xln2[r], [P["w0"], P["w1"]], n_seg=Fd, N=2 * Fd, out=hh[r])
                hip.ln_geglu_fwd(h0[r], h1[r], P["fln_w"], P["fln_b"], out=gln[r], mean=mean_f[r], rstd=rstd_f[r],
                                 q8=(gq[r], gs[r]) if fp8 else None)
                continue
            hip.gemm_nt(xln2[r], [P["w0"], P["w1"]], epilogue=hip.EPI_GEGLU, h0=h0[r] if keep else None, h1=h1[r] if keep else None, out=g[r])
            if has_fln:
                hip._check(L.op_layernorm_fwd(hip.ptr(g[r]), hip.ptr(P["fln_w"]), hip.ptr(P["fln_b"]), hip.ptr(gln[r]),
                                              hip.ptr(mean_f[r]) if keep else None, hip.ptr(rstd_f[r]) if keep else None, sg.rows, Fd, 1e-5, 0,
                                              hip.DT_BF16, None, hip.stream()), "op_layernorm_fwd")
        y2 = torch.empty(N, H, dtype=dt, device=dev) if keep and want_y else None
        out = torch.empty_like(x2)  # (with xmap: the full matrix)
        rs = [slice(sg.row0, sg.end) for sg in segs]
        if fp8:  # one fp8 launch per modality (the grouped persistent launch is bf16 only)
            for sg, P, r, ps in zip(segs, own, rs, pss):
                w2q, w2s = _fp8_weight(P["w2"])
                hip.gemm_nt_fp8(gq[r], gs[r], [w2q], [w2s], bias=P["b2"], epilogue=hip.EPI_RESID, resid=x2[r], gamma=shared["g2"], rowscale=ps,
                                rows_per_sample=sg.S, h0=y2[r] if y2 is not None else None, out=out[r])
            acts = dict(xln2=xln2, mean2=mean2, rstd2=rstd2, h0=h0, h1=h1, gln=gln, mean_f=mean_f, rstd_f=rstd_f, y2=y2) if keep else None
            return out, acts
        tabs = [xmap[r] for r in rs] if xmap is not None else None  # (row tables: every problem writes ITS rows of the shared full matrix)
        grouped = hip.gemm_nt_grouped([gln[r] for r in rs], [P["w2"] for P in own], biases=[P["b2"] for P in own],
                                      outs=[out[r] for r in rs] if tabs is None else [out] * nseg,
                                      epilogue=hip.EPI_RESID, h0s=[y2[r] for r in rs] if y2 is not None else None,
                                      resids=[x2[r] for r in rs] if tabs is None else [x2] * nseg, gammas=[shared["g2"]] * nseg,
                                      rowscales=list(pss), rows_per_sample=[sg.S for sg in segs], resid_rows=tabs)
        if grouped is None:  # shape outside the persistent kernel: one launch per modality
            for i, (sg, P, r, ps) in enumerate(zip(segs, own, rs, pss)):
                hip.gemm_nt(gln[r], [P["w2"]], [P["b2"]], epilogue=hip.EPI_RESID, resid=x2[r] if tabs is None else x2, gamma=shared["g2"],
                            rowscale=ps, rows_per_sample=sg.S, h0=y2[r] if y2 is not None else None, out=out[r] if tabs is None else out,
                            resid_rows=tabs[i] if tabs is not None else None)
        acts = dict(xln2=xln2, mean2=mean2, rstd2=rstd2, h0=h0, h1=h1, gln=gln, mean_f=mean_f, rstd_f=rstd_f, y2=y2) if keep else None
        return out, acts

    @staticmethod
    def forward(ctx, x2, segs, pss, save_acts, kept, *params):
        nseg = len(segs)
        Fd = params[3].shape[0]
        need_grad = any(ctx.needs_input_grad) and bool(int(save_acts) & 2)
        keep = bool(int(save_acts) & 1) and need_grad
        needs = ctx.needs_input_grad[5:]
        has_fln = params[5] is not None
        x_full = x2
        mapped = _row_tables(kept)  # (see AttnBranchFn.forward)
        if kept is not None and not mapped:  # the branch runs on the rows of the samples it keeps (segs / pss describe THOSE rows)
            x2 = hip.rows_gather(x_full, kept)
        H = x2.shape[1]
        N = kept.total if mapped else x2.shape[0]
        w2s, b2s = [params[3 + 6 * i + 4] for i in range(nseg)], [params[3 + 6 * i + 5] for i in range(nseg)]
        ctx.dg_fused = need_grad and dgamma_from_wgrad_ok([sg.rows for sg in segs], params[2], w2s, b2s, bool(needs[2]),
                                                           [bool(needs[3 + 6 * i + 4]) for i in range(nseg)])
        out, acts = FfnBranchMultiFn._compute(x2, segs, pss, params, keep, bool(needs[2]) and not ctx.dg_fused, grad=bool(int(save_acts) & 2),
                                              xmap=kept.rowmap() if mapped else None)
        if mapped:
            hip.rows_merge(x_full, None, kept, out=out)  # (the dropped samples' rows: copied)
        elif kept is not None:
            out = hip.rows_merge(x_full, out, kept)
        ctx.kept, ctx.mapped = kept, mapped
        ctx.segs, ctx.n_params, ctx.dims = segs, len(params), (N, H, Fd)
        ctx.direct = ()
        names = list(FFN_SHARED) + ["%s@%d" % (n, i) for i in range(nseg) for n in FFN_OWN]
        ctx.names = names
        if need_grad:
            _register_direct(ctx, names, params, needs)
        if keep:  # activations only a frozen parameter's gradient would read are not kept
            nd = dict(zip(names, needs))
            if not any(nd["w0@%d" % i] or nd["w1@%d" % i] for i in range(nseg)):
                acts["xln2"] = None
            if has_fln and not any(nd["w2@%d" % i] for i in range(nseg)):
                acts["gln"] = None
        ctx.cheap = keep and bool(int(save_acts) & 4)
        if ctx.cheap:  # (set_recompute_cheap)
            acts["xln2"] = None
            if _geglu_split(Fd, params[5]):
                acts["gln"] = None
        _save(ctx, keep, acts or {}, x2, *[ps for ps in pss], *params)
        ctx.n_ps = len(pss)
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, *rest = ctx.saved_tensors
        pss, rest = rest[:ctx.n_ps], rest[ctx.n_ps:]
        params, saved_acts = rest[:ctx.n_params], rest[ctx.n_params:]
        segs, names = ctx.segs, ctx.names
        nseg = len(segs)
        N, H, Fd = ctx.dims
        P = dict(zip(names, params))
        needs = {n: bool(ng) and q is not None for n, q, ng in zip(names, params, ctx.needs_input_grad[5:])}
        need_x = bool(ctx.needs_input_grad[0])
        kept, mapped = ctx.kept, ctx.mapped
        xmap = kept.rowmap() if mapped else None  # (mapped: x2 is the FULL matrix the forward read through the table)
        if ctx.act_names is not None:
            A = _restore(ctx, saved_acts, ("mean_f", "rstd_f", "y2", "gln", "xln2"))
            if ctx.cheap:
                if any(needs["%s@%d" % (n, i)] for i in range(nseg) for n in ("w0", "w1")):
                    A["xln2"] = hip.layernorm_fwd(x2, P["ln2_w"], P["ln2_b"], x_rows=xmap)[0]
                if A["gln"] is None and any(needs["w2@%d" % i] for i in range(nseg)):
                    A["gln"] = torch.empty(N, Fd, dtype=x2.dtype, device=x2.device)
                    for i, sg in enumerate(segs):
                        r = slice(sg.row0, sg.end)
                        hip.ln_geglu_fwd(A["h0"][r], A["h1"][r], P["fln_w@%d" % i], P["fln_b@%d" % i], out=A["gln"][r], want_stats=False)
        else:  # recompute (the reference's checkpoint_activations behaviour)
            _, A = FfnBranchMultiFn._compute(x2, segs, pss, params, True, not ctx.dg_fused, xmap=xmap)
        if not dout.is_contiguous():
            dout = dout.contiguous()
        dout_full = dout
        if kept is not None and not mapped:
            dout = hip.rows_gather(dout_full, kept)
        G = {}
        weight_grad, direct = _weight_grad_fn(ctx, G)
        has_fln = P["fln_w@0"] is not None
        dev, dt = dout.device, dout.dtype
        upstream = need_x or needs["ln2_w"] or needs["ln2_b"] or any(
            needs["%s@%d" % (n, i)] for i in range(nseg) for n in ("w0", "w1", "fln_w", "fln_b"))
        dh = torch.empty(N, 2 * Fd, dtype=dt, device=dev) if upstream else None
        dgln = torch.empty(N, Fd, dtype=dt, device=dev) if upstream else None
        colsets, dg_tmp = [], None
        fused = ctx.dg_fused  # gamma_2's gradient from the three down-projection weight gradients (no y2: dgamma_from_wgrad_ok)
        fp8b = FP8_FFN and FP8_FFN_DGRAD and H % 128 == 0 and Fd % 128 == 0
        (rowdot, rd_views), pairs = (_rowdot_slots(dev, [P["w2@%d" % i] for i in range(nseg)]), []) if fused else ((None, None), None)
        for i, sg in enumerate(segs):
            r = slice(sg.row0, sg.end)
            k = lambda n: "%s@%d" % (n, i)  # noqa: E731
            # residual branch: gamma_2 is shared (its gradient accumulates over the segments), the bias is the modality's own
            want_g, want_b = needs["g2"] and not fused, needs[k("b2")]
            acc_g, acc_b = "g2" in direct, k("b2") in direct
            tg = direct["g2"].grad if acc_g else (True if want_g else None)
            tb = direct[k("b2")].grad if acc_b else (True if want_b else None)
            if want_g and want_b and acc_g != acc_b:  # one in the flat buffer, one not: two temporaries, folded below
                tg = tb = True
                acc_g = acc_b = False
            g0 = None
            if fused:  # (g0 also switches the kernel to the un-scaled branch gradient: dy2 = ps * dout without gamma_2)
                g0 = torch.empty(H, dtype=torch.float32, device=dev)
                if P[k("b2")] is not None:
                    pairs.append((P[k("b2")], g0))
            dy2, dg_, db_ = hip.resid_bwd(dout_full if mapped else dout[r], A["y2"][r] if want_g else None, P["g2"], pss[i], sg.S,
                                          dgamma=tg if want_g else None, dbias=tb if want_b else None,
                                          accumulate=(acc_g and want_g) or (acc_b and want_b), g0=g0, dout_rows=xmap[r] if mapped else None)
            if want_g and not acc_g:
                dg_tmp = dg_.float() if dg_tmp is None else dg_tmp + dg_.float()
            if want_b:
                if acc_b:
                    _direct_grad_done(direct[k("b2")])
                else:
                    G[k("b2")] = db_
            if fused:  # (the hook rides on the last segment's problem: by then every (bias, g0) pair is listed)
                weight_grad(k("w2"), dy2, A["gln"][r], side=(P[k("w2")], rd_views[i], P["g2"]),
                            after=_gamma_finish_hook(rowdot, direct["g2"], pairs, True) if i == nseg - 1 else None)
            elif needs[k("w2")]:
                weight_grad(k("w2"), dy2, A["gln"][r])
            if not upstream:
                colsets.append(None)
                continue
            w2t = _transposed(P[k("w2")], scale=P["g2"] if fused else None)
            if fp8b:  # (opt-in, round 6) d LN_F(g) = dy W2 on e4m3 operands: the gradient rows and the transposed weight copy row-quantised
                hip.gemm_nt_fp8(*hip.quant_fp8_rows(dy2), *[[x] for x in _fp8_derived(w2t)], out=dgln[r])
            else:
                hip.gemm_nt(dy2, [w2t], out=dgln[r])
            pair = (k("w0"), k("w1"))
            order = _adjacent_grads(direct, pair) if needs[pair[0]] and needs[pair[1]] and Fd % 8 == 0 else None
            cols = tuple(order) if order else pair
            colsets.append(cols)
            dpart = {n: dh[r, j * Fd:(j + 1) * Fd] for j, n in enumerate(cols)}
            if has_fln:
                want = needs[k("fln_w")] or needs[k("fln_b")]
                (tw, tb2), acc = _targets(direct, k("fln_w"), k("fln_b")) if want else ((None, None), False)
                _, _, dw_, db2_ = hip.ln_geglu_bwd(dgln[r], A["h0"][r], A["h1"][r], P[k("fln_w")], A["mean_f"][r], A["rstd_f"][r], dw=tw,
                                                   db=tb2, accumulate=acc, need_wgrad=want, dh0=dpart[pair[0]], dh1=dpart[pair[1]])
                if want:
                    _finish(direct, G, (k("fln_w"), k("fln_b")), (dw_, db2_), acc)
            else:
                d0, d1 = hip.geglu_bwd(dgln[r], A["h0"][r], A["h1"][r])
                dpart[pair[0]].copy_(d0)
                dpart[pair[1]].copy_(d1)
            if order:
                wgrad_into(dh[r], A["xln2"][r], _span(direct, order), [direct[n] for n in order])
            else:
                for n in pair:
                    if needs[n]:
                        weight_grad(n, dpart[n], A["xln2"][r])
        if needs["g2"] and not fused:
            if "g2" in direct and dg_tmp is None:
                _direct_grad_done(direct["g2"])
            else:
                G["g2"] = dg_tmp.to(dt)
        dx = None
        if upstream and (need_x or needs["ln2_w"] or needs["ln2_b"]):
            dxln2 = torch.empty(N, H, dtype=dt, device=dev)
            rs = [slice(sg.row0, sg.end) for sg in segs]
            wts = [_transposed(tuple(P[n] for n in cols)) for cols in colsets]
            if fp8b:  # one fp8 launch per modality (the grouped persistent launch is bf16 only)
                for r, wt in zip(rs, wts):
                    hip.gemm_nt_fp8(*hip.quant_fp8_rows(dh[r]), *[[x] for x in _fp8_derived(wt)], out=dxln2[r])
            elif hip.gemm_nt_grouped([dh[r] for r in rs], wts, outs=[dxln2[r] for r in rs]) is None:
                for r, wt in zip(rs, wts):
                    hip.gemm_nt(dh[r], [wt], out=dxln2[r])
            want = needs["ln2_w"] or needs["ln2_b"]
            (tw, tb), acc = _targets(direct, "ln2_w", "ln2_b") if want else ((None, None), False)
            dx, dw_, db_ = hip.layernorm_bwd(dxln2, x2, P["ln2_w"], P["ln2_b"], A["mean2"], A["rstd2"], add=dout, dw=tw, db=tb,
                                             accumulate=acc, need_wgrad=want, dx=torch.empty_like(dout_full) if mapped else None, x_rows=xmap)
            if want:
                _finish(direct, G, ("ln2_w", "ln2_b"), (dw_, db_), acc)
        if dx is None and need_x:
            dx = dout_full
        elif dx is not None and mapped:
            hip.rows_merge(dout_full, None, kept, out=dx)  # (the dropped samples' rows: the gradient of the skip connection, copied)
        elif dx is not None and kept is not None:
            dx = hip.rows_merge(dout_full, dx, kept)
        grads = _return_grads(names, params, G, direct)
        return (dx, None, None, None, None, *grads)


def ffn_branch_multi(x2, segs, pss, shared, own, save_acts=True, kept=None):
    """shared: (ln2_w, ln2_b, g2); own: per segment (w0, w1, fln_w, fln_b, w2, b2); save_acts False: recompute in backward.
    kept (hip.KeptRows): as attn_branch_multi -- segs and pss describe the packed rows of the kept samples."""
    flat = list(shared)
    for o in own:
        flat += list(o)
    return FfnBranchMultiFn.apply(x2, segs, tuple(pss), _save_flags(save_acts), kept, *flat)


_SCALE_ROWS = {}


def kept_segments(kept, segs, device):
    """The packed-row view of a branch that skips its dropped samples: ([StreamSeg] over the rows of hip.KeptRows `kept`, an fp32
    vector of the drop-path multiplier 1/keep_prob long enough for every row- or sample-indexed use).  segs: the full-matrix
    segments (bias handles and key-padding rows are taken from them; key padding is gathered per kept sample)."""
    out = []
    for i, sg in enumerate(segs):
        key_pad = sg.key_pad.index_select(0, kept.kept_list(i)) if sg.key_pad is not None else None
        out.append(StreamSeg(sg.name, kept.n_kept[i], sg.S, kept.dst_row0[i], sg.bias, key_pad, pad=kept.dst_rows[i] - kept.n_kept[i] * sg.S))
    key = (device, kept.scale)
    vec = _SCALE_ROWS.get(key)
    if vec is None or vec.numel() < kept.total + 8:
        if len(_SCALE_ROWS) > 256:
            _SCALE_ROWS.clear()
        vec = _SCALE_ROWS[key] = torch.full((max(kept.full_rows, kept.total) + 8,), kept.scale, dtype=torch.float32, device=device)
    return out, vec


def _attn_backward(qkv, dattn, attn, lse, B, S, heads, scale, bias_img, biasT, key_pad, dbias_acc, bias_frag, dq, dk, dv):
    """Attention backward; dq / dk / dv: [N, H] column blocks of one packed [N, 3H] gradient matrix (any block order)."""
    H = heads * 64
    dev = qkv.device
    Spad = hip.attn_spad(S)
    delta = torch.empty(B, heads, Spad, dtype=torch.float32, device=dev)  # workspace of the call (rowsum(dO o O), from the dQ kernels)
    if attn.stride(0) != dattn.stride(0) or SEPARATE_DELTA:  # (the fused delta reads both with one row stride)
        hip._check(hip.lib().op_attn_bwd_delta(hip.ptr(dattn), hip.ptr(attn), dattn.stride(0), hip.ptr(delta), B, S, Spad, heads,
                                               hip.stream()), "op_attn_bwd_delta")
        attn = None
    hip.attn_bwd_launch(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], 3 * H, dattn, bias_img, biasT, key_pad, lse, delta, dq, dk,
                        dv, dq.stride(0), dbias_acc, B, S, Spad, heads, scale, bias_frag, out=attn)


def attn_branch(x, bias, key_pad, ps, heads, params, save_acts=False):
    """Single stream: x [B, S, H] contiguous; bias: RelPosBias-like handle or None; key_pad: uint8 [B, Spad] or None; ps: fp32 [B]
    drop-path multipliers or None."""
    B, S, H = x.shape
    seg = StreamSeg("x", B, S, 0, bias, key_pad)
    out = AttnBranchFn.apply(x.reshape(B * S, H), [seg], ps, S, heads, _save_flags(save_acts), None, bias.image if bias is not None else None, *params)
    return out.view(B, S, H)


def attn_branch_multi(x2, segs, ps_rows, heads, params, save_acts=False, kept=None):
    """Lock-step pass: x2 [sum rows, H] holds the rows of several streams (segs); ps_rows: fp32 [sum rows] per-row drop-path
    multipliers or None.  The attention-branch weights are modality-shared (transformer_layer.py:111-138), so LayerNorm, q|k|v,
    sub-LayerNorm and out-proj (and, in backward, their input- and weight-gradient GEMMs) are ONE launch over all rows.
    kept (hip.KeptRows): the branch runs only on the samples stochastic depth keeps -- segs and ps_rows then describe the PACKED
    rows (kept_segments), x2 stays the full matrix; rows of dropped samples pass through unchanged, forward and backward."""
    return AttnBranchFn.apply(x2, segs, ps_rows, 1, heads, _save_flags(save_acts), kept,
                              *[sg.bias.image if sg.bias is not None else None for sg in segs], *params)


def _save_flags(save_acts):
    """bit 0: keep the activations; bit 1: autograd is recording (inside Function.forward grad mode is always off); bit 2: keep them
    WITHOUT the four LayerNorm-type outputs backward can re-create (set_recompute_cheap).  The GeGLU form of
    the FFN is chosen by bit 1 -- not by which parameters happen to be trainable: a frozen branch (stage-2 pretraining) and a trainable
    one must give the same forward bits, and so must a checkpointed forward and its recomputation."""
    return int(bool(save_acts)) | (2 if torch.is_grad_enabled() else 0) | (4 if RECOMPUTE_CHEAP and save_acts else 0)


def ffn_branch(x, ps, params, save_acts=False):
    return FfnBranchFn.apply(x, ps, _save_flags(save_acts), *params)


def encoder_layer(x, bias, key_pad, ps1, ps2, heads, params, save_acts=False):
    """Whole single-modality layer; params in LAYER_PARAMS order."""
    n = len(ATTN_PARAMS)
    return ffn_branch(attn_branch(x, bias, key_pad, ps1, heads, params[:n], save_acts), ps2, params[n:], save_acts)


# --------------------------------------------------------------------------------------------------------------
# contrastive head
# --------------------------------------------------------------------------------------------------------------
class L2NormalizeFn(torch.autograd.Function):
    """F.normalize(x, dim=1) (one_peace_retrieval.py:112)."""

    @staticmethod
    def forward(ctx, x):
        y, inv = hip.l2norm_fwd(x.contiguous())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        return hip.l2norm_bwd(dy.contiguous(), y, inv)


def l2_normalize(x):
    return L2NormalizeFn.apply(x)


class InfoNCEFn(torch.autograd.Function):
    """image_text_pretrain_loss.py:164-185: both directions of scale * local @ all^T -> fp32 log-softmax -> NLL.

    a_local/b_local: [b, H] bf16 (with grad).  a_all/b_all: [n, H] bf16 gathered copies (no grad flows into them).
    scale: fp32 0-d tensor (logit_scale.exp()).  Returns (loss, a_hits, b_hits)."""

    @staticmethod
    def forward(ctx, a_local, b_local, a_all, b_all, scale, rank, label_smoothing):
        b = a_local.shape[0]
        t0 = rank * b
        scale_f = scale.detach().float().reshape(1)
        sims, rows = [], []
        for loc, allv in ((a_local, b_all), (b_local, a_all)):
            sim = gemm_any(loc.detach(), allv.detach(), out_f32=True, alpha=scale_f)
            loss_r, hit_r, dot_r = hip.infonce_rows(sim, t0, label_smoothing, gscale=0.5 / b, write_grad=True)
            sims.append(sim)  # now holds d loss / d sim
            rows.append((loss_r, hit_r, dot_r))
        loss = 0.5 * (rows[0][0].mean() + rows[1][0].mean())
        ctx.save_for_backward(a_local, b_local, a_all, b_all, scale_f, sims[0], sims[1], rows[0][2], rows[1][2])
        ctx.scale_dtype = scale.dtype
        hits_a, hits_b = rows[0][1].sum(), rows[1][1].sum()
        ctx.mark_non_differentiable(hits_a, hits_b)  # counters for logging, as under the reference's no-grad argmax
        return loss, hits_a, hits_b

    @staticmethod
    def backward(ctx, gl, _ga, _gb):
        a_local, b_local, a_all, b_all, scale_f, ds_ab, ds_ba, dot_ab, dot_ba = ctx.saved_tensors
        g = gl.float()
        coef = (g * scale_f).reshape(1)
        da = gemm_any(ds_ab.to(torch.bfloat16), hip.transpose(b_all.contiguous()), out_f32=True, alpha=coef)
        db = gemm_any(ds_ba.to(torch.bfloat16), hip.transpose(a_all.contiguous()), out_f32=True, alpha=coef)
        dscale = (g * (dot_ab.sum() + dot_ba.sum()) / scale_f).reshape(()).to(ctx.scale_dtype)
        return da.to(a_local.dtype), db.to(b_local.dtype), None, None, dscale, None, None


def info_nce(a_local, b_local, a_all, b_all, scale, rank=0, label_smoothing=0.0):
    return InfoNCEFn.apply(a_local, b_local, a_all, b_all, scale, rank, label_smoothing)


class DclFn(torch.autograd.Function):
    """Masked-token contrastive loss (image_text_pretrain_loss.py:187-208): rows = the MASKED student tokens, columns =
    every teacher token of the batch, target of row i = the teacher token at the same position.

    student [m, H] and teacher [n, H] are L2-normalised bf16; the caller has permuted the teacher rows so that the m
    masked positions come first in student order (softmax is permutation invariant), i.e. the target of row i is column
    i.  The [m, n] similarity matrix (22 k x 32 k at b=128) is never materialised: row blocks of `block` rows go
    through GEMM (fp32) -> op_infonce_rows (loss + d loss / d sim in place) -> GEMM with the teacher, and only the
    [m, H] student gradient is kept for backward.  The teacher carries no gradient (it is detached in the reference)."""

    @staticmethod
    def forward(ctx, student, teacher, scale, label_smoothing, block):
        m, H = student.shape
        teacher = teacher.detach().contiguous()
        teacher_t = hip.transpose(teacher)  # [H, n]: the NT operand of d student = d sim @ teacher
        alpha = torch.full((1,), float(scale), dtype=torch.float32, device=student.device)
        dstudent = torch.empty(m, H, dtype=torch.float32, device=student.device)
        total = torch.zeros((), dtype=torch.float32, device=student.device)
        for r0 in range(0, m, block):
            rows = student[r0:r0 + block].detach().contiguous()
            sim = gemm_any(rows, teacher, out_f32=True, alpha=alpha)
            loss_r, _, _ = hip.infonce_rows(sim, r0, label_smoothing, gscale=1.0 / m, write_grad=True)
            total += loss_r.sum()
            dstudent[r0:r0 + block] = gemm_any(sim.to(torch.bfloat16), teacher_t, out_f32=True, alpha=alpha)
        ctx.save_for_backward(dstudent)
        ctx.dtype = student.dtype
        return total / m

    @staticmethod
    def backward(ctx, g):
        (dstudent,) = ctx.saved_tensors
        return (dstudent * g.float()).to(ctx.dtype), None, None, None, None


def dcl_loss(student, teacher, scale, label_smoothing=0.0, block=4096):
    return DclFn.apply(student, teacher, scale, label_smoothing, block)
