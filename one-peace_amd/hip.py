"""ctypes binding of the C-ABI HIP library (include/onepeace_hip.h).

Every entry point takes raw device pointers + sizes + a hipStream_t; this module converts torch tensors,
passes ``torch.cuda.current_stream()`` and turns a non-zero return code into ``RuntimeError`` (so the
reference trainer's OOM / NaN handling paths still see Python exceptions).  There is NO CPU
fallback here: if the library is missing, loading fails loudly.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ONEPEACE_HIP_LIB") or os.path.join(_HERE, "lib", "libonepeace_hip.so")  # (override: instrumented builds of tools/)
_lib = None

DT_BF16, DT_F32 = 0, 1
EPI_BIAS, EPI_F32, EPI_GEGLU, EPI_RESID = 0, 1, 2, 3
PROF_GEMM, PROF_ATTN = 0, 1
ATTN_RESIDENT_MAX_S = 320  # sequences up to this length run the resident-K/V attention kernels (csrc/attention.hip: RES_MAX_S)

P = c_void_p
I64 = c_int64

# name -> (restype, argtypes); kept in the order of include/onepeace_hip.h
SIGNATURES = {
    "op_abi_version": (c_int, []),
    "op_last_error": (ctypes.c_char_p, []),
    "op_prof_enable": (c_int, [c_int]),
    "op_prof_reserve": (c_int, [c_int]),
    "op_prof_collect": (c_int, [P, P, P, c_int]),
    "op_layernorm_fwd": (c_int, [P, P, P, P, P, P, I64, I64, c_float, c_int, c_int, P, P]),
    "op_layernorm_bwd_workspace_bytes": (I64, [I64, I64]),
    "op_attn_bwd_dbias_slabs": (I64, [I64, I64, I64, I64]),
    "op_layernorm_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, I64, I64, c_int, c_int, c_int, P, P]),
    "op_gemm_plan": (c_int, [c_int64, c_int64, c_int64, c_int, c_int, c_int64, I64, P]),
    "op_gemm_nt": (c_int, [P, I64, P, P, P, I64, I64, P, P, P, P, I64, P, P, P, I64, P, P, I64, P, I64, I64, I64,
                           c_int, P, I64, I64, P, I64, P]),
    "op_gemm_nt_grouped": (c_int, [I64, P, P, I64, P, I64, P, P, I64, P, P, P, I64, P, P, P, I64, I64, c_int, I64, P, I64, P]),
    "op_gemm_tn": (c_int, [P, I64, P, I64, P, I64, I64, I64, I64, c_int, P, I64, I64, P]),
    "op_gemm_tn_grouped_counter_bytes": (I64, []),
    "op_gemm_tn_grouped_plan": (I64, [I64, P, P, P, I64, I64, P, I64]),
    "op_gemm_tn_grouped": (c_int, [I64, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, P]),
    "op_gemm_nt_batched": (c_int, [P, I64, I64, P, I64, I64, P, I64, P, I64, I64, I64, I64, I64, I64, P]),
    "op_audio_conv1_ln_gelu_fwd": (c_int, [P, I64, P, P, P, P, P, P, P, I64, I64, c_float, P]),
    "op_audio_conv1_ln_gelu_bwd_workspace_bytes": (I64, [I64]),
    "op_audio_conv1_ln_gelu_bwd": (c_int, [P, P, I64, P, P, P, P, P, P, P, P, P, P, P, I64, I64, c_int, P]),
    "op_transpose": (c_int, [P, P, I64, I64, I64, I64, P]),
    "op_transpose_scaled": (c_int, [P, P, I64, I64, I64, I64, P, P]),
    "op_transpose_batched": (c_int, [P, I64, I64, P]),
    "op_transpose_desc_bytes": (I64, []),
    "op_colsum_workspace_bytes": (I64, [I64]),
    "op_colsum_segments": (c_int, [P, P, P, P, P, I64, I64, I64, c_int, P]),
    "op_resid_bwd_workspace_bytes": (I64, [I64]),
    "op_resid_bwd": (c_int, [P, P, P, P, I64, P, P, P, P, P, I64, I64, c_int, P, P]),
    "op_gamma_grad_finish": (c_int, [P, I64, P, P, P, P, P, P, P, I64, c_int, P]),
    "op_ln_geglu_bwd": (c_int, [P, P, P, P, P, P, P, P, I64, I64, P, P, P, I64, I64, c_int, P]),
    "op_ln_geglu_fwd": (c_int, [P, P, I64, P, P, P, P, P, I64, I64, c_float, P]),
    "op_colsum": (c_int, [P, P, P, I64, P, P, P, I64, I64, c_int, c_int, P]),
    "op_geglu_bwd": (c_int, [P, P, P, P, P, I64, P]),
    "op_scale_rows": (c_int, [P, P, P, I64, P, I64, I64, P]),
    "op_l2norm_fwd": (c_int, [P, P, P, I64, I64, c_float, c_int, P]),
    "op_l2norm_bwd": (c_int, [P, P, P, P, I64, I64, c_int, P]),
    "op_infonce_rows": (c_int, [P, I64, I64, I64, I64, c_float, c_float, P, P, P, c_int, P]),
    "op_adamw_step": (c_int, [P, P, P, P, I64, c_float, c_float, c_float, c_float, c_float, I64, c_float, P, c_float, P]),
    "op_adamw_step_groups": (c_int, [P, P, P, P, I64, P, P, P, I64, c_float, c_float, c_float, c_float, I64, c_float, P, c_float, P]),
    "op_sqnorm": (c_int, [P, I64, P, P, P]),
    "op_relpos_bias_build": (c_int, [P, P, I64, P, I64, I64, I64, c_int, P]),
    "op_relpos_bias_bwd": (c_int, [P, P, I64, P, I64, I64, I64, P]),
    "op_relpos_bias_build_ids": (c_int, [P, P, I64, P, P, I64, I64, I64, I64, c_int, P]),
    "op_relpos_bias_bwd_ids": (c_int, [P, P, I64, P, P, I64, P, I64, I64, I64, I64, P]),
    "op_attn_fwd": (c_int, [P, P, P, I64, P, I64, P, P, P, I64, P, I64, I64, I64, I64, I64, I64, c_float, I64, P]),
    "op_attn_bias_frag_elems": (I64, [I64, I64]),
    "op_attn_bias_pack": (c_int, [P, P, I64, I64, I64, P]),
    "op_attn_bwd_delta": (c_int, [P, P, I64, P, I64, I64, I64, I64, P]),
    "op_attn_bwd": (c_int, [P, P, P, I64, P, P, I64, P, P, P, I64, P, P, P, P, P, P, I64, P, I64, I64, I64, I64, I64, c_float, I64, P]),
    "op_quant_fp8_rows": (c_int, [P, I64, P, I64, P, I64, I64, P]),
    "op_layernorm_fwd_q8": (c_int, [P, P, P, P, P, P, P, P, I64, I64, c_float, P]),
    "op_ln_geglu_fwd_q8": (c_int, [P, P, I64, P, P, P, P, P, P, P, I64, I64, c_float, P]),
    "op_gemm_nt_fp8": (c_int, [P, I64, P, P, P, I64, P, P, P, P, I64, P, P, P, I64, P, P, I64, I64, I64, I64, c_int, I64, P]),
    "op_rows_gather": (c_int, [P, P, P, I64, P, P, P, P, P, P, P, I64, I64, P]),
    "op_rows_merge": (c_int, [P, P, P, P, I64, P, P, P, P, P, P, P, I64, I64, P]),
    "op_rows_map": (c_int, [P, P, I64, P, P, P, P, P, P, P, I64, P]),
}


# libonepeace_probe.so (include/onepeace_probe.h): lane-map and power probes -- test / measurement infrastructure, not product ABI
PROBE_SIGNATURES = {
    "op_last_error": (ctypes.c_char_p, []),
    "op_probe_mfma16": (c_int, [P, P, P, c_int, P]),
    "op_probe_mfma32": (c_int, [P, P, P, c_int, P]),
    "op_probe_tr16": (c_int, [P, P, P, c_int, P]),
    "op_probe_glds": (c_int, [P, P, c_int, P, P]),
    "op_probe_mfma_f8": (c_int, [P, P, P, P, P, c_int, P]),
    "op_probe_mfma_rate": (c_int, [P, P, P, c_int, c_int, P]),
    "op_probe_lds_atomic": (c_int, [P, P, c_int, c_int, c_int, P]),
    "op_probe_occupy": (c_int, [P, ctypes.c_longlong, c_int, ctypes.c_longlong, P, P]),
}


class Tuning:
    """Kernel-flavour selection for tests and tools.  The C library keeps no tuning state: every GEMM / attention entry point
    takes a per-call `tune` word (include/onepeace_hip.h), and this host-side object is where the Python binding keeps the
    values it passes.  Production code never touches it (all zeros = the defaults)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.tile_mode = 0       # 0 auto, 1 force 128x128, 2 force 256x256
        self.fullline = 2        # flavour of the 256x256 NT kernel: 0 BK = 32, 1 eight-wave full-line, 2 auto, 3 four-wave full-line
        self.tail_rows = 1       # 0 off, 1 default, 2 whenever it saves a round, 3 always
        self.gm = 0              # M-tiles per L2 group (0 auto)
        self.force_splits = 0    # forced K-split count of small problems (tools)
        self.glds = 1            # 1 LDS-DMA staging, 0 register-staged operands
        self.sched = int(os.environ.get("ONEPEACE_TUNE_SCHED", "0"))  # four-wave NT launches: 0 auto (persistent gemm256p_kernel for K <= 2048), 1 / 3 gemm256v_kernel, 6 gemm256p_kernel for every single problem (A/B), 7 one tile per workgroup: gemm256v_kernel, grouped launches unrolled into the tile list (distributed.share_cus_with_collectives)
        self.fp8_small = 0       # op_gemm_nt_fp8: 1 = keep the 128 x 128 kernel (tests, A/B)
        self.merge_dbias = 1     # attention backward: 1 merged dQ + dBias kernel, 0 separate kernels
        self.resident = 1        # attention forward: bit 0 resident kernels on; bits 1-2 ablations (tools)
        self.attn_waves = 0      # resident forward kernel: waves per workgroup (0 = production rule; tests / A-B timing)
        self.attn_pers = 1       # attention forward: 1 persistent kernel for 193 ... 257 tokens (round 4), 0 the resident one
        self.dbias_chunks_r2 = 0  # merged dQ + dBias kernel: round 2's batch-chunk rule (A/B timing)
        self.dkdv_keys = 0       # dK/dV kernel: 0 auto, 1 = 128 keys per workgroup, 2 = 64 (A/B timing)
        self.dbias_chunks = 0    # merged dQ + dBias kernel: forced number of batch chunks (0 = the library's rule; A/B timing)
        self.attn_pers_dkdv = 1  # attention backward: persistent dK / dV kernel behind the persistent dQ kernel (round 4), 0 the rounds 1-3 kernel
        self.attn_lone_keys = 1  # dK/dV: a trailing block of <= 16 keys (the 257th token) split over the waves by queries (round 4), 0 one wave
        self.attn_pers_bwd = 1   # attention backward: 1 persistent dQ (+ dBias) kernel for 193 ... 257 tokens (round 4), 0 rounds 1-3

    def gemm(self):
        fl = {2: 0, 0: 1, 1: 2, 3: 3}.get(self.fullline, 0)
        tr = 0 if self.tail_rows == 1 else self.tail_rows + 1
        return (self.tile_mode | fl << 2 | tr << 4 | (self.gm & 31) << 7 | (self.force_splits & 15) << 15
                | (0 if self.glds else 1) << 19 | (self.sched & 7) << 20)

    def attn_fwd(self):
        return (0 if self.resident & 1 else 1) | ((self.resident >> 1) & 3) << 1 | (self.attn_waves & 15) << 3 | (0 if self.attn_pers else 128)

    def attn_bwd(self):
        return ((0 if self.merge_dbias else 1) | (2 if self.dbias_chunks_r2 else 0) | (self.dkdv_keys & 3) << 2 | (self.dbias_chunks & 63) << 4
                | (0 if self.attn_pers_bwd else 1024) | (0 if self.attn_lone_keys else 2048) | (0 if self.attn_pers_dkdv else 4096))


TUNE = Tuning()


class _LibProxy:
    """The ctypes library plus the historical op_*_set_* knob functions, now pure Python on `TUNE` (tools/ and tests/ written
    against the round-1 ABI keep working; the shared library no longer exports or stores any knob)."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        return getattr(self._cdll, name)

    @staticmethod
    def op_gemm_set_tile(mode):
        old = TUNE.tile_mode
        if mode >= 60:
            TUNE.force_splits = mode - 60
        elif mode >= 50:
            TUNE.tail_rows = mode - 50
        elif mode >= 40:
            TUNE.gm = mode - 40
        elif mode >= 20:
            if 0 <= mode - 20 <= 3:  # other values were ignored by the round-1 C knob as well
                TUNE.fullline = mode - 20
        elif mode >= 10:
            pass  # (rounds 1-4: timing ablations of the BK = 32 kernel, removed in round 5)
        else:
            TUNE.tile_mode = mode
        return old

    @staticmethod
    def op_gemm_set_staging(glds):
        old, TUNE.glds = TUNE.glds, 1 if glds else 0
        return old

    @staticmethod
    def op_attn_set_merge_dbias(on):
        old, TUNE.merge_dbias = TUNE.merge_dbias, 1 if on else 0
        return old

    @staticmethod
    def op_attn_set_resident(mode):
        old, TUNE.resident = TUNE.resident, int(mode)
        return old


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "%s not found: build it with `python one-peace_amd/build.py` (hipcc, gfx950). "
                "There is no CPU fallback for the HIP path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = _LibProxy(L)
    return _lib


_probe_lib = None


def probe_lib():
    """The probe library (tests/test_probes_gpu.py, bench.py's power-limited MFMA rate); built next to the product library."""
    global _probe_lib
    if _probe_lib is None:
        path = os.path.join(os.path.dirname(LIB_PATH), "libonepeace_probe.so")
        if not os.path.exists(path):
            raise RuntimeError("%s not found: build it with `python one-peace_amd/build.py`" % path)
        L = ctypes.CDLL(path)
        for name, (res, args) in PROBE_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _probe_lib = L
    return _probe_lib


def _check_probe(rc, what):
    if rc != 0:
        msg = probe_lib().op_last_error()
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))


def available():
    return os.path.exists(LIB_PATH) and torch.cuda.is_available()


def _check(rc, what):
    if rc != 0:
        msg = lib().op_last_error()
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t):
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float32:
        return DT_F32
    raise TypeError("unsupported dtype %s" % t.dtype)


def _req(t, name, dtype=None):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA(HIP) tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))


# ---------------------------------------------------------------------------------------------------
# thin tensor-level wrappers (no autograd here; see ops.py)
# ---------------------------------------------------------------------------------------------------
def layernorm_fwd(x2d, w, b, eps=1e-5, gelu=False, want_stats=True, q8=False, x_rows=None):
    """q8=True (bf16, no GELU): returns (y, mean, rstd, (fp8 bytes, row scales)) -- the output also row-quantised to e4m3, bit-identical
    to quant_fp8_rows(y), without a pass of its own.  x_rows (int32 [rows], KeptRows.rowmap): the input rows are rows x_rows[r] of the
    larger matrix x2d (< 0: a row of zeros); the outputs have x_rows.numel() rows."""
    _req(x2d, "x")
    rows, cols = x2d.shape
    if x_rows is not None:
        assert not q8 and x_rows.dtype == torch.int32 and x_rows.is_contiguous()
        rows = x_rows.numel()
    y = torch.empty(rows, cols, dtype=x2d.dtype, device=x2d.device)
    mean = rstd = None
    if want_stats:
        mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    if q8:
        assert not gelu and x2d.dtype == torch.bfloat16
        q = torch.empty(rows, cols, dtype=torch.uint8, device=x2d.device)
        qs = torch.empty(rows, dtype=torch.float32, device=x2d.device)
        _check(lib().op_layernorm_fwd_q8(ptr(x2d), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), ptr(q), ptr(qs), rows, cols, eps, stream()),
               "op_layernorm_fwd_q8")
        return y, mean, rstd, (q, qs)
    _check(lib().op_layernorm_fwd(ptr(x2d), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps,
                                  int(gelu), _dt(x2d), ptr(x_rows), stream()), "op_layernorm_fwd")
    return y, mean, rstd


_ws_cache = {}


def workspace(nbytes, device, tag="ws"):
    """Grow-only scratch buffer per (device, tag, stream); caller-owned memory as the C ABI requires."""
    key = (device, tag, torch.cuda.current_stream(device).cuda_stream)  # per stream: launches on different streams may overlap
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def layernorm_bwd(dy, x, w, b, mean, rstd, gelu=False, need_wgrad=True, dw=None, db=None, accumulate=False, add=None,
                  dx=None, x_rows=None):
    """x_rows (int32 [rows of dy]): x, add and dx are rows x_rows[r] of larger matrices; dx must be given (dx = add: in place, the
    rows no entry names keep `add`)."""
    rows, cols = x.shape
    if x_rows is not None:
        assert dx is not None and dx.shape == x.shape and (add is None or add.shape == x.shape) and x_rows.dtype == torch.int32
        rows = x_rows.numel()
        assert dy.shape[0] == rows
    if dx is None:
        dx = torch.empty_like(x)
    ws = None
    if need_wgrad and w is not None:
        if dw is None:
            dw = torch.empty_like(w)
            db = torch.empty_like(w)
            accumulate = False
        ws = workspace(lib().op_layernorm_bwd_workspace_bytes(rows, cols), x.device, "ln")
    else:
        dw = db = None
    _check(lib().op_layernorm_bwd(ptr(dy), ptr(x), ptr(w), ptr(b), ptr(mean), ptr(rstd), ptr(add), ptr(dx), ptr(dw), ptr(db),
                                  ptr(ws), rows, cols, int(gelu), int(accumulate), _dt(x), ptr(x_rows), stream()), "op_layernorm_bwd")
    return dx, dw, db


SPLITK_WS_BYTES = 768 << 20
GEMM_ALGO_BYTES = [0, 0]  # [bytes, launches]: operands read once + outputs written once (bench.py's roofline.traffic yardstick)


def gemm_nt(A, Bs, biases=None, out=None, epilogue=EPI_BIAS, n_seg=0, h0=None, h1=None, resid=None, gamma=None,
            rowscale=None, rows_per_sample=0, alpha=None, N=None, ldc=None, splitk=True, resid_rows=None):
    """C[M,N] = A[M,K] @ cat(Bs)[N,K]^T with the fused epilogue.  Bs: list of 1-3 [n_seg,K] weights (GeGLU: [W0, W1]).
    resid_rows (int32 [M], residual epilogue): resid and out are FULL matrices, row m of the launch reads / writes their row
    resid_rows[m] (< 0: dropped); h0 stays [M, N]."""
    M, K = A.shape
    if resid_rows is not None:
        assert epilogue == EPI_RESID and out is not None and resid is not None and out.shape == resid.shape and resid_rows.numel() == M
        assert resid_rows.dtype == torch.int32 and resid_rows.is_contiguous()
    Bs = list(Bs) + [None] * (3 - len(Bs))
    biases = list(biases or []) + [None] * (3 - len(biases or []))
    if epilogue == EPI_GEGLU:
        Nn = Bs[0].shape[0]
    else:
        Nn = N if N is not None else sum(b.shape[0] for b in Bs if b is not None)
        if n_seg == 0:
            n_seg = Bs[0].shape[0]
    if out is None:
        out = torch.empty(M, Nn, dtype=torch.float32 if epilogue == EPI_F32 else torch.bfloat16, device=A.device)
    ldc_ = ldc if ldc is not None else out.stride(0)
    ws, ws_bytes = None, 0
    # fp32 scratch for split-K: bias-free plain launches with few tiles and a long K (weight gradients, dgrads), and the
    # small latency-bound launches (a few M-tiles: the leftover rows of a tail-rows split, batch-1 feature extraction)
    # whose bias / residual epilogue the fold kernel applies.  The library decides; handing the scratch over costs nothing.
    if splitk and epilogue in (EPI_BIAS, EPI_RESID) and (K >= 2048 or M <= 1024):
        # at most 8 fp32 slabs of the output (the planner's limit); small launches (feature extraction under hipGraph
        # capture, where every capture stream gets its own scratch) then pin megabytes, not the training-size buffer
        ws = workspace(min(SPLITK_WS_BYTES, 32 * M * Nn), A.device, "gemm_splitk")
        ws_bytes = ws.numel()
    outs = 1 + (2 if h1 is not None else (1 if h0 is not None else 0))
    GEMM_ALGO_BYTES[0] += 2 * (M * K + Nn * K * (2 if epilogue == EPI_GEGLU else 1) + (M * Nn if resid is not None else 0)) \
        + out.element_size() * M * Nn * outs
    GEMM_ALGO_BYTES[1] += 1
    _check(lib().op_gemm_nt(ptr(A), A.stride(0), ptr(Bs[0]), ptr(Bs[1]), ptr(Bs[2]), Bs[0].stride(0), n_seg,
                            ptr(biases[0]), ptr(biases[1]), ptr(biases[2]), ptr(out), ldc_, ptr(h0), ptr(h1),
                            ptr(resid), resid.stride(0) if resid is not None else 0, ptr(gamma), ptr(rowscale),
                            rows_per_sample, ptr(alpha), M, Nn, K, epilogue, ptr(ws), ws_bytes, TUNE.gemm(), ptr(resid_rows),
                            out.shape[0] if resid_rows is not None else 0, stream()), "op_gemm_nt")
    return out


def _ptr_array(items, n):
    arr = (c_void_p * n)()
    for i, t in enumerate(items or ()):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def gemm_nt_grouped(As, Ws, biases=None, outs=None, epilogue=EPI_BIAS, h0s=None, h1s=None, resids=None, gammas=None,
                    rowscales=None, rows_per_sample=None, resid_rows=None):
    """(resid_rows: per problem an int32 row table, see gemm_nt -- outs / resids are then the full matrices, one row count for all.)
    One launch for up to three problems out_p = epilogue(A_p @ W_p^T) with a common N, K (the per-modality FFNs of a layer).
    As: [M_p, K] tensors with a common row stride; Ws: per problem one weight, or (wi_0, wi_1) for the GeGLU epilogue.
    Returns the list of outputs, or None when the shape does not qualify for the persistent kernel (caller falls back)."""
    n = len(As)
    K = As[0].shape[1]
    pairs = [w if isinstance(w, (tuple, list)) else (w, None) for w in Ws]
    N = pairs[0][0].shape[0]
    if outs is None:
        outs = [torch.empty(a.shape[0], N, dtype=torch.bfloat16, device=a.device) for a in As]
    lda, ldb, ldc = As[0].stride(0), pairs[0][0].stride(0), outs[0].stride(0)
    ldr = resids[0].stride(0) if resids and resids[0] is not None else 0
    if any(a.stride(0) != lda or a.shape[1] != K for a in As) or any(w[0].stride(0) != ldb or w[0].shape != (N, K) for w in pairs) \
            or any(o.stride(0) != ldc for o in outs) or (ldr and any(r.stride(0) != ldr for r in resids)):
        return None
    flatB, flatb = [], []
    for i, (w0, w1) in enumerate(pairs):
        flatB += [w0, w1]
        flatb += [biases[i] if biases else None, None]
    Ms = (c_int64 * n)(*[a.shape[0] for a in As])
    rps = (c_int64 * n)(*[(rows_per_sample[i] if rows_per_sample else 0) for i in range(n)])
    for i, a in enumerate(As):
        outs_n = 1 + (2 if h1s and h1s[i] is not None else (1 if h0s and h0s[i] is not None else 0))
        GEMM_ALGO_BYTES[0] += 2 * (a.shape[0] * K + N * K * (2 if epilogue == EPI_GEGLU else 1)
                                   + (a.shape[0] * N if resids and resids[i] is not None else 0)) + 2 * a.shape[0] * N * outs_n
    GEMM_ALGO_BYTES[1] += 1
    rc = lib().op_gemm_nt_grouped(n, _ptr_array(As, n), Ms, lda, _ptr_array(flatB, 2 * n), ldb, _ptr_array(flatb, 2 * n),
                                  _ptr_array(outs, n), ldc, _ptr_array(h0s, n), _ptr_array(h1s, n), _ptr_array(resids, n), ldr,
                                  _ptr_array(gammas, n), _ptr_array(rowscales, n), rps, N, K, epilogue, TUNE.gemm(),
                                  _ptr_array(resid_rows, n) if resid_rows is not None else None,
                                  outs[0].shape[0] if resid_rows is not None else 0, stream())
    if rc == -95:
        return None
    _check(rc, "op_gemm_nt_grouped")
    return outs


def quant_fp8_rows(x2d, out=None):
    """bf16 [rows, cols] -> (fp8 e4m3 bytes [rows, cols] as uint8, fp32 row scales [rows]) with x ~= q * scale[row].
    out: an earlier result to overwrite in place (derived weight copies keep their addresses across optimiser steps)."""
    rows, cols = x2d.shape
    if out is not None:
        q, scale = out
    else:
        q = torch.empty(rows, cols, dtype=torch.uint8, device=x2d.device)
        scale = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    _check(lib().op_quant_fp8_rows(ptr(x2d), x2d.stride(0), ptr(q), q.stride(0), ptr(scale), rows, cols, stream()), "op_quant_fp8_rows")
    return q, scale


def gemm_nt_fp8(A8, sa, B8s, sbs, bias=None, out=None, epilogue=EPI_BIAS, h0=None, h1=None, resid=None, gamma=None, rowscale=None,
                rows_per_sample=0):
    """C[M,N] bf16 = epilogue((A8 @ B8^T) * sa[:, None] * sb[None, :]) on the fp8 MFMA path; A8 [M,K] / B8 [N,K] uint8 (e4m3),
    per-row fp32 scales.  B8s / sbs: [W] or, for the GeGLU epilogue, [W0, W1]."""
    M, K = A8.shape
    N = B8s[0].shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=A8.device)
    B1, sb1 = (B8s[1], sbs[1]) if len(B8s) > 1 else (None, None)
    _check(lib().op_gemm_nt_fp8(ptr(A8), A8.stride(0), ptr(sa), ptr(B8s[0]), ptr(B1), B8s[0].stride(0), ptr(sbs[0]), ptr(sb1),
                                ptr(bias), ptr(out), out.stride(0), ptr(h0), ptr(h1), ptr(resid),
                                resid.stride(0) if resid is not None else 0, ptr(gamma), ptr(rowscale), rows_per_sample, M, N, K,
                                epilogue, int(TUNE.fp8_small), stream()), "op_gemm_nt_fp8")
    return out


def gemm_plan(M, N, K, epilogue=EPI_BIAS, has_bias=True, workspace_bytes=SPLITK_WS_BYTES):
    """(tile, K-splits, epilogue-in-fold-kernel, tail rows) op_gemm_nt would use; a host-only query (works without a GPU)."""
    out = (ctypes.c_int * 4)()
    _check(lib().op_gemm_plan(M, N, K, epilogue, int(has_bias), workspace_bytes, TUNE.gemm(), ctypes.cast(out, P)), "op_gemm_plan")
    return out[0], out[1], bool(out[2]), out[3]


def gemm_tn_supported(K, M, N, lda, ldb):
    return (K % 64 == 0 and M % 8 == 0 and N % 8 == 0 and lda % 8 == 0 and ldb % 8 == 0 and M >= 8 and N >= 8
            and 31 * lda + M < (1 << 30) and 31 * ldb + N < (1 << 30))


def gemm_nt_batched(A, W, bias, out, rows, K):
    """out[z] = A_z W[z]^T (+ bias[z]) for z < G as ONE launch.  A: [G, *, lda] whose batch z starts at A[z] and whose ROWS are read with
    stride A.stride(1) as K-wide (possibly overlapping) patches; W [G, N, K] contiguous; bias [G, N] or None; out [G, rows, N]."""
    G, N = W.shape[0], W.shape[1]
    assert W.is_contiguous() and out.is_contiguous() and out.shape == (G, rows, N) and A.stride(2) == 1
    GEMM_ALGO_BYTES[0] += 2 * G * (rows * K + N * K + rows * N)
    GEMM_ALGO_BYTES[1] += 1
    _check(lib().op_gemm_nt_batched(ptr(A), A.stride(1), A.stride(0), ptr(W), K, N * K, ptr(bias), N if bias is not None else 0, ptr(out), N,
                                    rows * N, rows, N, K, G, stream()), "op_gemm_nt_batched")
    return out


def audio_conv1_ln_gelu_fwd(wav, stride, w0, b0, lnw, lnb, rows, eps, slack_rows=0):
    """GELU(LN(conv(wav))) of the feature extractor's first block straight from the flat waveform: y [rows + slack_rows, C] (the slack
    rows zero: the next block's strided patch view reads past the last row), mean, rstd [rows]."""
    C = w0.shape[0]
    assert w0.is_contiguous() and w0.numel() == C * 10 and wav.is_contiguous() and wav.numel() >= stride * (rows - 1) + 10
    y = torch.empty(rows + slack_rows, C, dtype=torch.bfloat16, device=wav.device)
    if slack_rows:
        y[rows:].zero_()
    mean = torch.empty(rows, dtype=torch.float32, device=wav.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=wav.device)
    _check(lib().op_audio_conv1_ln_gelu_fwd(ptr(wav), stride, ptr(w0), ptr(b0), ptr(lnw), ptr(lnb), ptr(y), ptr(mean), ptr(rstd), rows, C, eps,
                                            stream()), "op_audio_conv1_ln_gelu_fwd")
    return y, mean, rstd


def audio_conv1_ln_gelu_bwd(dy, wav, stride, w0, b0, lnw, lnb, mean, rstd):
    """Parameter gradients of audio_conv1_ln_gelu_fwd (bf16): dw0 [C, 10], db0 [C] | None, dlnw [C] | None, dlnb [C] | None."""
    rows, C = dy.shape
    mk = lambda ref, shape: torch.empty(shape, dtype=torch.bfloat16, device=dy.device) if ref is not None else None  # noqa: E731
    dw0, db0, dlw, dlb = torch.empty(C, 10, dtype=torch.bfloat16, device=dy.device), mk(b0, C), mk(lnw, C), mk(lnb, C)
    ws = workspace(lib().op_audio_conv1_ln_gelu_bwd_workspace_bytes(C), dy.device, "audio_conv1")
    _check(lib().op_audio_conv1_ln_gelu_bwd(ptr(dy), ptr(wav), stride, ptr(w0), ptr(b0), ptr(lnw), ptr(lnb), ptr(mean), ptr(rstd), ptr(dw0), ptr(db0),
                                            ptr(dlw), ptr(dlb), ptr(ws), rows, C, 0, stream()), "op_audio_conv1_ln_gelu_bwd")
    return dw0, db0, dlw, dlb


def gemm_tn_grouped_plan(sizes, workgroups=256, tune=0):
    """[(M, N, K)] -> [(queue, problem, tile_m, tile_n)] in draw order: op_gemm_tn_grouped's schedule (host-only query)."""
    n = len(sizes)
    cap = sum(((m + 255) // 256) * ((nn + 255) // 256) for m, nn, _ in sizes)
    out = (ctypes.c_int32 * (4 * cap))()
    arr = lambda j: (c_int64 * n)(*[q[j] for q in sizes])  # noqa: E731
    cnt = lib().op_gemm_tn_grouped_plan(n, arr(0), arr(1), arr(2), int(workgroups), int(tune), ctypes.cast(out, P), cap)
    if cnt < 0:
        raise RuntimeError("op_gemm_tn_grouped_plan failed (%d)" % cnt)
    return [tuple(out[4 * i:4 * i + 4]) for i in range(cnt)]


def gemm_tn(A_km, B_kn, out=None, accumulate=False, splitk=True):
    """C[M,N] (+)= A_km^T @ B_kn with A_km [K, M], B_kn [K, N] (row-major, last dim contiguous): dW = dy^T x without copies.
    splitk=False: no scratch is handed over, every output tile runs its whole K (tests: the grouped launch's yardstick)."""
    K, M = A_km.shape
    N = B_kn.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=A_km.device)
        accumulate = False
    ws = workspace(SPLITK_WS_BYTES, A_km.device, "gemm_splitk") if splitk else None
    GEMM_ALGO_BYTES[0] += 2 * (K * M + K * N + M * N * (2 if accumulate else 1))
    GEMM_ALGO_BYTES[1] += 1
    _check(lib().op_gemm_tn(ptr(A_km), A_km.stride(0), ptr(B_kn), B_kn.stride(0), ptr(out), out.stride(0), M, N, K,
                            int(accumulate), ptr(ws), ws.numel() if ws is not None else 0, TUNE.gemm(), stream()), "op_gemm_tn")
    return out


_tn_counters = {}
TN_GROUP_MAX = 16


def gemm_tn_grouped(problems, tune=0):
    """ONE persistent launch for up to 16 weight-gradient GEMMs, no split-K (csrc/gemm.hip: gemm256w_tn_grouped_kernel).
    problems: [(A_km [K, M], B_kn [K, N], out [M, N] bf16, accumulate[, (W [M, N] bf16, rowdot fp32 [N / 128, M][, rscale bf16 [M]])])].
    Returns False (nothing launched) when a problem does not qualify for the transpose-read kernel -- the caller then runs gemm_tn per
    problem.  (W, rowdot): rowdot[s][m] = sum over the 128 columns n of slot s of W[m][n] * (this launch's fp32 product)[m][n] (written,
    every entry once); rscale: out[m] += rscale[m] * product[m] while rowdot sums the unscaled product -- see gamma_grad_finish."""
    n = len(problems)
    dev = problems[0][0].device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ctr = _tn_counters.get(key)
    if ctr is None:  # zeroed once; every launch re-arms it
        ctr = _tn_counters[key] = torch.zeros(int(lib().op_gemm_tn_grouped_counter_bytes()), dtype=torch.uint8, device=dev)
    arr = lambda vals: (c_int64 * n)(*vals)  # noqa: E731
    As, Bs, Cs = [q[0] for q in problems], [q[1] for q in problems], [q[2] for q in problems]
    acc = (ctypes.c_int32 * n)(*[int(bool(q[3])) for q in problems])
    side = [tuple(q[4]) + (None,) * (3 - len(q[4])) if len(q) > 4 and q[4] is not None else (None, None, None) for q in problems]
    Ws, Rs, Ss = [x[0] for x in side], [x[1] for x in side], [x[2] for x in side]
    has_side = any(r is not None for r in Rs)
    for sc, r, a, b in zip(Ss, Rs, As, Bs):
        assert sc is None or (sc.dtype == torch.bfloat16 and sc.is_contiguous() and sc.numel() == a.shape[1])
        assert r is None or (r.dtype == torch.float32 and r.is_contiguous() and r.shape == (b.shape[1] // 128, a.shape[1])), "rowdot: fp32 [N / 128, M]"
    rc = lib().op_gemm_tn_grouped(n, _ptr_array(As, n), arr([a.stride(0) for a in As]), _ptr_array(Bs, n), arr([b.stride(0) for b in Bs]),
                                  _ptr_array(Cs, n), arr([c.stride(0) for c in Cs]), arr([a.shape[1] for a in As]),
                                  arr([b.shape[1] for b in Bs]), arr([a.shape[0] for a in As]), acc,
                                  _ptr_array(Ws, n) if has_side else None, arr([w.stride(0) if w is not None else 0 for w in Ws]) if has_side else None,
                                  _ptr_array(Rs, n) if has_side else None, _ptr_array(Ss, n) if any(x is not None for x in Ss) else None,
                                  ptr(ctr), int(tune), stream())
    if rc == -95:
        return False
    _check(rc, "op_gemm_tn_grouped")
    for a, b, c, ac in [q[:4] for q in problems]:
        K, M = a.shape
        GEMM_ALGO_BYTES[0] += 2 * (K * M + K * b.shape[1] + M * b.shape[1] * (2 if ac else 1))
    GEMM_ALGO_BYTES[1] += 1
    return True


def transpose(x2d, out=None, scale=None):
    """out[c][r] = x2d[r][c] (* scale[r]: bf16 [rows], the product rounded to bf16)."""
    rows, cols = x2d.shape
    if out is None:
        out = torch.empty(cols, rows, dtype=x2d.dtype, device=x2d.device)
    assert scale is None or (scale.dtype == torch.bfloat16 and scale.is_contiguous() and scale.numel() == rows)
    _check(lib().op_transpose_scaled(ptr(x2d), ptr(out), rows, cols, x2d.stride(0), out.stride(0), ptr(scale), stream()), "op_transpose")
    return out


def transpose_table(jobs, device):
    """jobs: [(src [rows, cols] (last dim contiguous), dst view [cols, rows] (last dim contiguous)[, scale bf16 [rows] or None])] ->
    (device table, total tiles) for transpose_batched; the table stays valid as long as the tensors keep their storage."""
    import struct
    assert lib().op_transpose_desc_bytes() == 56
    raw, tile0 = b"", 0
    for job in jobs:
        src, dst = job[:2]
        scale = job[2] if len(job) > 2 else None
        rows, cols = src.shape
        tx, ty = (cols + 63) // 64, (rows + 63) // 64
        raw += struct.pack("<QQiiqqiiQ", src.data_ptr(), dst.data_ptr(), rows, cols, src.stride(0), dst.stride(0), tile0, tx,
                           scale.data_ptr() if scale is not None else 0)
        tile0 += tx * ty
    table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    return table, tile0


def transpose_batched(table, n, total_tiles):
    _check(lib().op_transpose_batched(ptr(table), n, total_tiles, stream()), "op_transpose_batched")


def colsum(x, y=None, rowscale=None, rows_per_sample=0, mul=None, out=None, accumulate=False, out_dtype=torch.bfloat16):
    M, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=out_dtype, device=x.device)
        accumulate = False
    ws = workspace(lib().op_colsum_workspace_bytes(N), x.device, "colsum")
    _check(lib().op_colsum(ptr(x), ptr(y), ptr(rowscale), rows_per_sample, ptr(mul), ptr(out), ptr(ws), M, N,
                           int(accumulate), _dt(out), stream()), "op_colsum")
    return out


def colsum_segments(x, seg_cols, outs=None, accumulate=False):
    """Per-segment column sums of x [M, n_seg*seg_cols]; outs: list of bf16 [seg_cols] targets / None (skip) per segment."""
    M, N = x.shape
    n_seg = N // seg_cols
    if outs is None:
        outs = [torch.empty(seg_cols, dtype=x.dtype, device=x.device) for _ in range(n_seg)]
        accumulate = False
    o = list(outs) + [None] * (3 - len(outs))
    ws = workspace(lib().op_colsum_workspace_bytes(N), x.device, "colsum")
    _check(lib().op_colsum_segments(ptr(x), ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(ws), M, n_seg, seg_cols, int(accumulate),
                                    stream()), "op_colsum_segments")
    return outs


def resid_bwd(dout, y=None, gamma=None, rowscale=None, rows_per_sample=0, dgamma=None, dbias=None, accumulate=False, g0=None,
              dout_rows=None):
    """dbranch = rowscale*gamma*dout plus the column reductions dgamma / dbias in one pass.  dgamma / dbias: True
    (allocate), a bf16 [N] tensor (write or, with accumulate, add into it) or None (skip).  g0: fp32 [N] that receives
    sum_m rowscale*dout (dbias without the gamma factor: gamma_grad_finish's operand) -- and dbranch is then rowscale*dout WITHOUT
    gamma (the weight-gradient launch's rscale and the gamma-scaled transposed weight carry it)."""
    M, N = dout.shape
    if dout_rows is not None:  # dout: a larger matrix, row m of the pass = its row dout_rows[m] (< 0: zeros); the result has M rows
        assert dout_rows.dtype == torch.int32 and dout_rows.is_contiguous() and dout.is_contiguous()
        M = dout_rows.numel()
    out = torch.empty(M, N, dtype=dout.dtype, device=dout.device)
    if dgamma is True:
        dgamma = torch.empty(N, dtype=dout.dtype, device=dout.device)
    if dbias is True:
        dbias = torch.empty(N, dtype=dout.dtype, device=dout.device)
    ws = None
    if dgamma is not None or dbias is not None or g0 is not None:
        ws = workspace(lib().op_resid_bwd_workspace_bytes(N), dout.device, "resid")
    assert g0 is None or (g0.dtype == torch.float32 and g0.is_contiguous() and g0.numel() == N)
    _check(lib().op_resid_bwd(ptr(dout), ptr(y if dgamma is not None else None), ptr(gamma), ptr(rowscale), rows_per_sample,
                              ptr(out), ptr(dgamma), ptr(dbias), ptr(g0), ptr(ws), M, N, int(accumulate), ptr(dout_rows), stream()),
           "op_resid_bwd")
    return out, dgamma, dbias


def gamma_grad_finish(rowdot, pairs, dgamma, accumulate):
    """dgamma (+)= sum_s rowdot[s] + sum_i b_i * g0_i.  rowdot: fp32 [slots, N], the partial row dots of gemm_tn_grouped over the
    un-gamma-scaled gradient (several weight sets that share gamma: their slots one after the other); pairs: up to three (bias bf16 [N]
    or None, g0 fp32 [N])."""
    assert rowdot.dtype == torch.float32 and rowdot.dim() == 2 and rowdot.is_contiguous() and len(pairs) <= 3 and dgamma.dtype == torch.bfloat16
    assert rowdot.shape[1] == dgamma.numel()
    flat = []
    for b, g0 in list(pairs) + [(None, None)] * (3 - len(pairs)):
        flat += [ptr(b), ptr(g0)]
    _check(lib().op_gamma_grad_finish(ptr(rowdot), rowdot.shape[0], *flat, ptr(dgamma), rowdot.shape[1], int(accumulate), stream()),
           "op_gamma_grad_finish")
    return dgamma


def ln_geglu_bwd(dy, h0, h1, w, mean, rstd, dw=None, db=None, accumulate=False, need_wgrad=True, dh0=None, dh1=None):
    """Backward of LayerNorm_F(gelu(h0)*h1): returns dh0, dh1, dw, db (dw/db allocated unless given; None with need_wgrad=False).
    dh0 / dh1 may be given as column blocks of one wider matrix (same row stride, last dim contiguous)."""
    rows, cols = h0.shape
    if dh0 is None:
        dh0, dh1 = torch.empty_like(h0), torch.empty_like(h1)
    assert dh0.stride(0) == dh1.stride(0) and dh0.stride(1) == 1 and dh1.stride(1) == 1
    assert h0.stride(0) == h1.stride(0) and h0.stride(1) == 1 and h1.stride(1) == 1
    ws = None
    if need_wgrad:
        if dw is None:
            dw, db, accumulate = torch.empty_like(w), torch.empty_like(w), False
        ws = workspace(lib().op_layernorm_bwd_workspace_bytes(rows, cols), h0.device, "ln")
    else:
        dw = db = None
    _check(lib().op_ln_geglu_bwd(ptr(dy), ptr(h0), ptr(h1), ptr(w), ptr(mean), ptr(rstd), ptr(dh0), ptr(dh1), dh0.stride(0),
                                 h0.stride(0), ptr(dw), ptr(db), ptr(ws), rows, cols, int(accumulate), stream()), "op_ln_geglu_bwd")
    return dh0, dh1, dw, db


def ln_geglu_fwd(h0, h1, w, b, eps=1e-5, want_stats=True, out=None, mean=None, rstd=None, q8=None):
    """LayerNorm_F(bf16(gelu(h0) * h1)); h0 / h1 may be column blocks of one wider matrix.  Returns y, mean, rstd.
    q8 = (fp8 bytes [rows, cols], scales [rows]): also filled with the row-quantised output (see layernorm_fwd)."""
    rows, cols = h0.shape
    assert h0.stride(0) == h1.stride(0) and h0.stride(1) == 1 and h1.stride(1) == 1
    y = out if out is not None else torch.empty(rows, cols, dtype=h0.dtype, device=h0.device)
    if want_stats and mean is None:
        mean = torch.empty(rows, dtype=torch.float32, device=h0.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=h0.device)
    if q8 is not None:
        assert q8[0].is_contiguous() and q8[0].shape == (rows, cols) and y.is_contiguous()
        _check(lib().op_ln_geglu_fwd_q8(ptr(h0), ptr(h1), h0.stride(0), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), ptr(q8[0]), ptr(q8[1]),
                                        rows, cols, eps, stream()), "op_ln_geglu_fwd_q8")
        return y, mean, rstd
    _check(lib().op_ln_geglu_fwd(ptr(h0), ptr(h1), h0.stride(0), ptr(w), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps, stream()),
           "op_ln_geglu_fwd")
    return y, mean, rstd


def geglu_bwd(dg, h0, h1):
    dh0, dh1 = torch.empty_like(h0), torch.empty_like(h1)
    _check(lib().op_geglu_bwd(ptr(dg), ptr(h0), ptr(h1), ptr(dh0), ptr(dh1), dg.numel(), stream()), "op_geglu_bwd")
    return dh0, dh1


def scale_rows(dout, gamma=None, rowscale=None, rows_per_sample=0):
    M, N = dout.shape
    out = torch.empty_like(dout)
    _check(lib().op_scale_rows(ptr(dout), ptr(gamma), ptr(rowscale), rows_per_sample, ptr(out), M, N, stream()),
           "op_scale_rows")
    return out


def l2norm_fwd(x, out_dtype=torch.bfloat16, eps=1e-12):
    rows, cols = x.shape
    y = torch.empty(rows, cols, dtype=out_dtype, device=x.device)
    inv = torch.empty(rows, dtype=torch.float32, device=x.device)
    _check(lib().op_l2norm_fwd(ptr(x), ptr(y), ptr(inv), rows, cols, eps, _dt(y), stream()), "op_l2norm_fwd")
    return y, inv


def l2norm_bwd(dy, y, inv):
    rows, cols = y.shape
    dx = torch.empty(rows, cols, dtype=torch.bfloat16, device=y.device)
    _check(lib().op_l2norm_bwd(ptr(dy), ptr(y), ptr(inv), ptr(dx), rows, cols, _dt(y), stream()), "op_l2norm_bwd")
    return dx


def infonce_rows(sim, target0, label_smoothing=0.0, gscale=1.0, write_grad=True):
    rows, n = sim.shape
    dev = sim.device
    loss = torch.empty(rows, dtype=torch.float32, device=dev)
    hit = torch.empty(rows, dtype=torch.float32, device=dev)
    dot = torch.empty(rows, dtype=torch.float32, device=dev)
    _check(lib().op_infonce_rows(ptr(sim), rows, n, sim.stride(0), target0, label_smoothing, gscale, ptr(loss), ptr(hit),
                                 ptr(dot), int(write_grad), stream()), "op_infonce_rows")
    return loss, hit, dot


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, grad_sqnorm=None, clip_norm=0.0):
    _check(lib().op_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                               grad_scale, ptr(grad_sqnorm), clip_norm, stream()), "op_adamw_step")


def adamw_step_groups(p, g, m, v, group_end8, group_lr_scale, group_wd, lr, beta1, beta2, eps, step, grad_scale=1.0,
                      grad_sqnorm=None, clip_norm=0.0):
    """One launch over the whole flat buffer; group tables are device tensors (int64 ends in 8-element vectors, fp32 scales)."""
    _check(lib().op_adamw_step_groups(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(group_end8), ptr(group_lr_scale),
                                      ptr(group_wd), group_end8.numel(), lr, beta1, beta2, eps, step, grad_scale, ptr(grad_sqnorm),
                                      clip_norm, stream()), "op_adamw_step_groups")


def sqnorm(x, out=None):
    """Sum of squares of a flat bf16 tensor -> fp32 device scalar [1]."""
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws = workspace(4096, x.device, "sqnorm")
    _check(lib().op_sqnorm(ptr(x), x.numel(), ptr(ws), ptr(out), stream()), "op_sqnorm")
    return out


def relpos_bias_build(table, bucket_i32, S, Spad, transposed=False):
    """[heads][S][Spad] bf16 image of table[bucket]; transposed=True gives out[h][key][query]."""
    heads = table.shape[1]
    out = torch.empty(heads, S, Spad, dtype=torch.bfloat16, device=table.device)
    _check(lib().op_relpos_bias_build(ptr(table), ptr(bucket_i32), bucket_i32.stride(0), ptr(out), heads, S, Spad,
                                      int(transposed), stream()), "op_relpos_bias_build")
    return out


def relpos_bias_bwd(dbias_f32, bucket_i32, num_rel, S, Spad):
    heads = dbias_f32.shape[0]
    dtable = torch.zeros(num_rel, heads, dtype=torch.float32, device=dbias_f32.device)
    _check(lib().op_relpos_bias_bwd(ptr(dbias_f32), ptr(bucket_i32), bucket_i32.stride(0), ptr(dtable), heads, S, Spad,
                                    stream()), "op_relpos_bias_bwd")
    return dtable


def relpos_bias_build_ids(table, bucket_i32, ids_i32, Kpad, transposed=False):
    """Per-sample images [B, heads, K, Kpad] of table[bucket[ids[b, i], ids[b, j]]] (see op_relpos_bias_build_ids)."""
    B, K = ids_i32.shape
    heads = table.shape[1]
    out = torch.empty(B, heads, K, Kpad, dtype=torch.bfloat16, device=table.device)
    _check(lib().op_relpos_bias_build_ids(ptr(table), ptr(bucket_i32), bucket_i32.stride(0), ptr(ids_i32), ptr(out), B, heads, K, Kpad,
                                          int(transposed), stream()), "op_relpos_bias_build_ids")
    return out


def relpos_bias_bwd_ids(dbias_f32, bucket_i32, ids_i32, num_rel):
    B, heads, K, Kpad = dbias_f32.shape
    dtable = torch.zeros(num_rel, heads, dtype=torch.float32, device=dbias_f32.device)
    Sfull = bucket_i32.shape[0]
    dense = torch.zeros(heads, Sfull, Sfull, dtype=torch.float32, device=dbias_f32.device)
    _check(lib().op_relpos_bias_bwd_ids(ptr(dbias_f32), ptr(bucket_i32), bucket_i32.stride(0), ptr(ids_i32), ptr(dense), Sfull,
                                        ptr(dtable), B, heads, K, Kpad, stream()), "op_relpos_bias_bwd_ids")
    return dtable


def attn_spad(S):
    """Padded key/query extent shared by the bias images, key-pad masks, lse and delta rows."""
    return ((S + 127) // 128) * 128


def _bias_bstride(bias):
    """bias [heads, S, Spad]: one image for all samples (stride 0); [B, heads, S, Spad]: one image per sample."""
    return bias.stride(0) if bias is not None and bias.dim() == 4 else 0


def attn_bias_pack(image, S):
    """Row-major bias image(s) [heads, S, Spad] or [B, heads, S, Spad] -> the fragment-major layout the resident forward
    kernel adds with the matrix pipe (flat bf16 tensor; see op_attn_bias_pack)."""
    n_img = image.numel() // (image.shape[-2] * image.shape[-1])
    out = torch.empty(lib().op_attn_bias_frag_elems(n_img, S), dtype=torch.bfloat16, device=image.device)
    _check(lib().op_attn_bias_pack(ptr(image), ptr(out), n_img, S, image.shape[-1], stream()), "op_attn_bias_pack")
    return out


def attn_fwd(q, k, v, ld, B, S, heads, scale, bias=None, key_pad=None, Spad=0, out=None, want_lse=True, bias_frag=None):
    """q, k, v: bf16 views into [B*S, ld] rows (head h at columns h*64..); returns out [B*S, heads*64] and
    lse [B, heads, Spad] (fp32, natural log; entries >= S are unspecified).  bias_frag: attn_bias_pack(bias, S) -- with it
    sequences of up to 320 keys take the resident-K/V kernel (without it a biased call runs the streaming kernel)."""
    dev = q.device
    H = heads * 64
    Spad = Spad or attn_spad(S)
    if out is None:
        out = torch.empty(B * S, H, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, heads, Spad, dtype=torch.float32, device=dev) if want_lse else None
    _check(lib().op_attn_fwd(ptr(q), ptr(k), ptr(v), ld, ptr(bias), _bias_bstride(bias), ptr(bias_frag), ptr(key_pad), ptr(out), out.stride(0),
                             ptr(lse), Spad, B, S, Spad, heads, 64, scale, TUNE.attn_fwd(), stream()), "op_attn_fwd")
    return out, lse


def attn_bwd(q, k, v, ld, dout, out, lse, B, S, heads, scale, bias=None, biasT=None, key_pad=None, Spad=0, dqkv=None,
             want_dbias=False, bias_frag=None):
    """Returns dqkv [B*S, 3H] (dq | dk | dv packed like a fused projection output) and dbias fp32: [heads,S,Spad] for a
    shared bias image, [B,heads,S,Spad] when bias / biasT hold one image per sample."""
    dev = q.device
    H = heads * 64
    Spad = Spad or attn_spad(S)
    delta = torch.empty(B, heads, Spad, dtype=torch.float32, device=dev)  # workspace: filled by the dQ kernels (out is passed)
    assert out.stride(0) == dout.stride(0)
    if dqkv is None:
        dqkv = torch.empty(B * S, 3 * H, dtype=torch.bfloat16, device=dev)
    per_sample = bias is not None and bias.dim() == 4
    dbias = attn_dbias_buffer(B, S, heads, Spad, dev, per_sample) if want_dbias else None
    dq, dk, dv = dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:]
    attn_bwd_launch(q, k, v, ld, dout, bias, biasT, key_pad, lse, delta, dq, dk, dv, dqkv.stride(0), dbias, B, S, Spad, heads, scale,
                    bias_frag, out=out)
    if dbias is None:
        return dqkv, None
    return dqkv, (dbias if per_sample else dbias.sum(0))


def attn_bwd_launch(q, k, v, ld, dout, bias, biasT, key_pad, lse, delta, dq, dk, dv, ldg, dbias, B, S, Spad, heads, scale,
                    bias_frag=None, out=None):
    """out (the forward output, row stride as dout): delta is then only a workspace, computed inside the call."""
    _check(lib().op_attn_bwd(ptr(q), ptr(k), ptr(v), ld, ptr(dout), ptr(out), dout.stride(0), ptr(bias), ptr(biasT), ptr(bias_frag),
                             _bias_bstride(bias),
                             ptr(key_pad), ptr(lse), ptr(delta), ptr(dq), ptr(dk), ptr(dv), ldg, ptr(dbias), B, S, Spad, heads,
                             64, scale, TUNE.attn_bwd(), stream()), "op_attn_bwd")


def attn_dbias_buffer(B, S, heads, Spad, device, per_sample=False):
    """Zeroed fp32 [slabs, heads, S, Spad] accumulator for op_attn_bwd's dbias: a shared bias image gets one slab per
    batch chunk (the gradient is the sum over dim 0), per-sample images one slab per sample (slab b = gradient of image b)."""
    slabs = B if per_sample else lib().op_attn_bwd_dbias_slabs(B, S, heads, TUNE.attn_bwd())
    return torch.zeros(slabs, heads, S, Spad, dtype=torch.float32, device=device)


class profile_kernels:
    """Context manager: record HIP-event pairs around gemm/attention launches on their launch stream."""

    def __enter__(self):
        lib().op_prof_enable(1)
        return self

    def __exit__(self, *exc):
        lib().op_prof_enable(0)

    @staticmethod
    def collect(n=4):
        ms = (c_double * n)()
        cnt = (c_int64 * n)()
        work = (c_double * n)()
        lib().op_prof_collect(ms, cnt, work, n)
        return [dict(ms=ms[i], count=cnt[i], work=work[i]) for i in range(n)]


class KeptRows:
    """The samples ONE residual branch of ONE layer keeps under stochastic depth, as the row packing op_rows_gather / op_rows_merge
    work with.  segs: [(src_row0, S, n_samples, kept sample numbers ascending)] per segment of the packed activation matrix;
    `lists`: the device int32 tensor that holds, from `base` on, for every segment its kept list followed by its inverse list
    (position among the kept samples or -1) -- built for the whole stack at once by `pack_kept_lists` (one host-to-device copy per
    step).  Every segment's packed rows are rounded up to a multiple of `pad` (zero rows: the weight-gradient kernels want
    K % 64 == 0, whole 256-row tiles avoid tail launches)."""

    def __init__(self, segs, lists, base, full_rows, scale, pad=256):
        n = len(segs)
        assert 1 <= n <= 4
        self.lists, self.full_rows, self.scale, self.nseg = lists, int(full_rows), float(scale), n
        self.S = [int(s[1]) for s in segs]
        self.n_samples = [int(s[2]) for s in segs]
        self.n_kept = [len(s[3]) for s in segs]
        self.src_row0 = [int(s[0]) for s in segs]
        self.dst_rows = [-(-(k * S) // pad) * pad for k, S in zip(self.n_kept, self.S)]
        self.dst_row0, self.off_kept, self.off_inv = [], [], []
        r, o = 0, int(base)
        for k, ns, dr in zip(self.n_kept, self.n_samples, self.dst_rows):
            self.dst_row0.append(r)
            self.off_kept.append(o)
            self.off_inv.append(o + k)
            r, o = r + dr, o + k + ns
        self.total, self.list_end = r, o
        arr = lambda v: (c_int64 * n)(*v)  # noqa: E731
        self._c = tuple(arr(v) for v in (self.src_row0, self.dst_row0, self.S, self.n_kept, self.dst_rows, self.n_samples))
        self._c_kept, self._c_inv = arr(self.off_kept), arr(self.off_inv)

    def kept_list(self, i):
        """Device int32 view: the kept sample numbers of segment i (e.g. to index_select the key-padding rows)."""
        return self.lists[self.off_kept[i]:self.off_kept[i] + self.n_kept[i]]

    def rowmap(self):
        """Device int32 [total]: the row of the full matrix behind every packed row (-1: a surplus row of a rounded-up segment) -- the
        table the kernels that read / write THROUGH the packing take (op_rows_map; built once per branch and step, on first use)."""
        m = getattr(self, "_rowmap", None)
        if m is None:
            m = torch.empty(self.total, dtype=torch.int32, device=self.lists.device)
            c = self._c
            _check(lib().op_rows_map(ptr(m), ptr(self.lists), self.nseg, c[0], c[1], c[2], c[3], c[4], c[5], self._c_kept, self.total,
                                     stream()), "op_rows_map")
            self._rowmap = m
        return m


def pack_kept_lists(plans):
    """plans: [[(src_row0, S, n_samples, kept list)] per segment] per branch -> (int32 CPU tensor with all kept / inverse lists, the
    offset of each plan's first entry).  Host only."""
    out, bases = [], []
    for segs in plans:
        bases.append(len(out))
        for _, _, ns, kept in segs:
            inv = [-1] * ns
            for j, smp in enumerate(kept):
                inv[smp] = j
            out.extend(int(v) for v in kept)
            out.extend(inv)
    return torch.tensor(out if out else [0], dtype=torch.int32), bases


def rows_gather(src, kr):
    """Pack the rows of the kept samples: [kr.total, cols] (zero rows where a segment is rounded up)."""
    assert src.dim() == 2 and src.is_contiguous() and src.shape[0] == kr.full_rows and src.dtype == torch.bfloat16
    dst = torch.empty(kr.total, src.shape[1], dtype=src.dtype, device=src.device)
    c = kr._c
    _check(lib().op_rows_gather(ptr(src), ptr(dst), ptr(kr.lists), kr.nseg, c[0], c[1], c[2], c[3], c[4], c[5], kr._c_kept, kr.total,
                                src.shape[1], stream()), "op_rows_gather")
    return dst


def rows_merge(base, upd, kr, out=None):
    """base with the rows of the kept samples replaced by the packed rows `upd`; out=base: in place.  upd=None (out given, not base):
    only the rows of the DROPPED samples are copied into out (the kept rows were written through KeptRows.rowmap)."""
    assert base.dim() == 2 and base.is_contiguous() and base.shape[0] == kr.full_rows and base.dtype == torch.bfloat16
    if upd is None:
        assert out is not None and out is not base and out.shape == base.shape and out.is_contiguous()
    else:
        assert upd.is_contiguous() and upd.shape[0] == kr.total and upd.dtype == torch.bfloat16 and base.shape[1] == upd.shape[1]
    if out is None:
        out = torch.empty_like(base)
    c = kr._c
    _check(lib().op_rows_merge(ptr(base), ptr(upd), ptr(out), ptr(kr.lists), kr.nseg, c[0], c[1], c[2], c[3], c[4], c[5], kr._c_inv,
                               kr.full_rows, base.shape[1], stream()), "op_rows_merge")
    return out


def mfma_rate_probe(seconds=1.0, waves_per_cu=8, data="normal", device=None):
    """The bf16 MFMA rate this GPU sustains with nothing else going on (op_probe_mfma_rate: register operands only), after the
    package has settled at its power limit: {"tflops", "mhz" (shader clock over the loop, mean / min / max), "ms", "workgroups"}.
    `data`: "normal" (random normal operands, what a training step multiplies), "zeros" (no toggling: the clock stays at its maximum)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    wgs = cus * max(1, (int(waves_per_cu) + 3) // 4)
    g = torch.Generator().manual_seed(1)
    img = (torch.randn(8 * 64 * 8, generator=g) * 0.5 if data == "normal" else torch.zeros(8 * 64 * 8)).to(device=device, dtype=torch.bfloat16)
    out = torch.empty(wgs * 256, dtype=torch.float32, device=device)
    clk = torch.zeros(wgs * 2, dtype=torch.int64, device=device)
    iters, ms = 20000, 0.0
    for leg in range(3):  # calibrate, settle (clock and power follow the load with a delay), report
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(device))
        _check_probe(probe_lib().op_probe_mfma_rate(ptr(img), ptr(out), ptr(clk), wgs, iters, stream()), "op_probe_mfma_rate")
        e1.record(torch.cuda.current_stream(device))
        e1.synchronize()
        ms = e0.elapsed_time(e1)
        if leg == 0:
            iters = max(1, int(iters * seconds * 1e3 / ms))
    c = clk.view(wgs, 2).double().cpu()
    mhz = c[:, 0] / (c[:, 1] * 0.01)
    return {"tflops": wgs * 4 * iters * 64 * 16384.0 / (ms * 1e-3) / 1e12, "ms": ms, "workgroups": wgs,
            "mhz": {"mean": float(mhz.mean()), "min": float(mhz.min()), "max": float(mhz.max())}}
