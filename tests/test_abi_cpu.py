"""The C-ABI library loads without a GPU and exports every symbol include/onepeace_hip.h declares; the ctypes table in
one-peace_amd/hip.py covers the same set with matching arity; the HIP path refuses to run without a device."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls(header="onepeace_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(op_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return decls


@pytest.fixture(scope="module")
def lib_path():
    import importlib.util
    spec = importlib.util.spec_from_file_location("onepeace_build", os.path.join(ROOT, "one-peace_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(verbose=False)


def test_library_exports_every_declared_symbol(lib_path):
    decls = _header_decls()
    assert len(decls) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = sorted(set(decls) - exported)
    assert not missing, "declared in the header but not exported: %s" % missing


def test_ctypes_table_matches_header(lib_path):
    from one_peace_amd import hip
    decls = _header_decls()
    for name, nargs in decls.items():
        assert name in hip.SIGNATURES, "%s missing from hip.SIGNATURES" % name
        assert len(hip.SIGNATURES[name][1]) == nargs, "%s: header has %d args, ctypes table %d" % (
            name, nargs, len(hip.SIGNATURES[name][1]))
    L = hip.lib()
    assert L.op_abi_version() == 9  # 9: row tables (round 6); 8 and earlier: 2: per-call tune words instead of process-wide knobs; 3: grouped GEMM, ldd of op_ln_geglu_bwd (round 3); 4: op_gemm_tn_grouped (round 4); 5: op_probe_mfma_rate; 6: probes in their own library (round 5); 7: row-dot side product of op_gemm_tn_grouped, g0 of op_resid_bwd, op_gamma_grad_finish; 8: rscale / op_transpose_scaled: the layer-scale gradient without a division (round 6)


def test_probe_library_is_separate_from_the_product_library(lib_path):
    """Round 5 (hygiene): the hardware / power probes are test and measurement infrastructure -- their own shared
    library and header; the product library exports none of them."""
    from one_peace_amd import hip
    probe_path = os.path.join(os.path.dirname(lib_path), "libonepeace_probe.so")
    decls = _header_decls("onepeace_probe.h")
    assert len(decls) >= 6 and all(n.startswith("op_probe_") or n == "op_last_error" for n in decls), decls

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True, check=True).stdout
        return {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert not (set(decls) - exported(probe_path))
    assert not [n for n in exported(lib_path) if n.startswith("op_probe_")]
    assert not [n for n in _header_decls() if n.startswith("op_probe_")]
    for name, nargs in decls.items():
        assert name in hip.PROBE_SIGNATURES and len(hip.PROBE_SIGNATURES[name][1]) == nargs, name
    hip.probe_lib()


def test_no_silent_cpu_fallback():
    """On a host without a GPU the HIP wrappers must fail loudly, never compute on the CPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from one_peace_amd import hip, ops
    x = torch.randn(4, 64)
    assert not ops.hip_eligible(x)
    with pytest.raises(RuntimeError):
        hip.layernorm_fwd(x.to(torch.bfloat16), None, None)
    from one_peace_amd.distributed import FlatParameters
    from one_peace_amd.optim import FusedAdamW
    lin = torch.nn.Linear(8, 8).to(torch.bfloat16)
    with pytest.raises(RuntimeError):
        FusedAdamW(FlatParameters(lin))


def test_gemm_launch_planner_decisions(lib_path):
    """op_gemm_plan is a host-only query of op_gemm_nt's launch decision (tile, K-splits, epilogue fold, tail-rows split):
    the decisions the measurements in profiles/r1_gemm_experiments.md and profiles/r1_gemm_small_m.txt led to."""
    from one_peace_amd import hip
    H, F = 1536, 6144
    M = 128 * 257  # the headline batch: 128 image sequences of 257 tokens
    # training shapes: 256 x 256 tiles, no K-split, the 128 leftover rows (M % 256) as a second launch when that saves a round
    for N, K, epi in ((3 * H, H, hip.EPI_BIAS), (H, H, hip.EPI_RESID), (H, F, hip.EPI_RESID)):
        tile, splits, fold, tail = hip.gemm_plan(M, N, K, epi)
        assert (tile, splits, fold, tail) == (256, 1, False, 128), (N, K, epi)
    assert hip.gemm_plan(M, F, H, hip.EPI_GEGLU)[:2] == (256, 1)
    assert hip.gemm_plan(64 * 256, 3 * H, H)[3] == 0                      # M a multiple of 256: nothing to split off
    # batch-1 feature extraction (M = 257): 128 x 128 tiles; K split over the idle CUs, epilogue in the fold kernel
    tile, splits, fold, tail = hip.gemm_plan(257, H, H)
    assert tile == 128 and splits in (3, 4) and fold and tail == 0
    tile, splits, fold, tail = hip.gemm_plan(257, H, F, hip.EPI_RESID)
    assert tile == 128 and splits >= 6 and fold
    assert hip.gemm_plan(257, 3 * H, H)[:2] == (128, 1)                  # 108 tiles: a split does not pay (measured)
    assert hip.gemm_plan(257, F, H, hip.EPI_GEGLU)[:3] == (128, 1, False)  # two accumulators: no fold for GeGLU
    assert hip.gemm_plan(257, H, F, hip.EPI_RESID, workspace_bytes=0)[1] == 1   # no scratch, no split
    # bias-free launches (dgrads of the tail rows) split K without the fold epilogue; beyond 1024 rows bias epilogues never fold
    tile, splits, fold, tail = hip.gemm_plan(128, H, F, hip.EPI_BIAS, has_bias=False)
    assert tile == 128 and splits > 1 and not fold
    assert hip.gemm_plan(2056, H, F, hip.EPI_RESID)[1] == 1
    with pytest.raises(RuntimeError):
        hip.gemm_plan(257, H, 100)  # K must be a multiple of 64


LAYER_WGRADS = [  # (M = out features, N = in features, K = tokens) of the weight gradients of one lock-step layer at b = 128
    (4608, 1536, 73088), (1536, 1536, 73088),                      # q|k|v, out-proj over all rows
    (12288, 1536, 32896), (1536, 6144, 32896),                     # image wi_0|wi_1, wo
    (12288, 1536, 32000), (1536, 6144, 32000),                     # audio
    (12288, 1536, 8192), (1536, 6144, 8192),                       # text
]


def test_grouped_weight_gradient_schedule_covers_every_tile_once(lib_path):
    """op_gemm_tn_grouped_plan (host-only): every 256 x 256 output tile of every problem sits in exactly one of the eight queues,
    problems in order of decreasing K inside a queue; also for ragged sizes, a grid wider than 8 in both directions, forced
    workgroup counts (3 / 8 / 37: the GPU tests' stealing and re-arming cases) and round 4's form of the queues (tune bit 10)."""
    from one_peace_amd import hip
    for sizes in (LAYER_WGRADS, [(264, 520, 128), (2304, 2560, 64), (8, 8, 64)], [(1536, 1536, 4096)] * 12):
        for nwg, tune in ((256, 0), (3, 0), (8, 0), (37, 0), (256, 1 << 10)):
            plan = hip.gemm_tn_grouped_plan(sizes, workgroups=nwg, tune=tune)
            tiles = {(p, tm, tn) for _, p, tm, tn in plan}
            want = {(p, tm, tn) for p, (M, N, _) in enumerate(sizes) for tm in range((M + 255) // 256) for tn in range((N + 255) // 256)}
            assert len(plan) == len(want) and tiles == want
            for x in range(8):
                ks = [sizes[p][2] for q, p, _, _ in plan if q == x]
                assert ks == sorted(ks, reverse=True)


def test_grouped_weight_gradient_schedule_deals_waves_that_share_their_panels(lib_path):
    """Round 5 (VERDICT r4 #1a: the grouped launch fetched 16.9 GB for 4.5 GB of operands): what the 32 workgroups of an XCD run at the
    same time -- 32 consecutive draws of its queue -- is a rectangle of ONE problem's tile grid (or of two problems of the same K):
    all tiles of a wave have the same K (they finish together and the next wave starts together), and a full wave touches at most
    13 operand panels (6 of the narrow + 6-7 of the wide operand) instead of 64.  The only waves that mix K are the partial last
    waves of the K classes (at most one per class).  Round 4's round-robin deal of six-tile groups put groups of different problems
    side by side: on average 19 panels per 32 tiles, and their start times drifted apart."""
    from one_peace_amd import hip
    plan = hip.gemm_tn_grouped_plan(LAYER_WGRADS)
    assert len(plan) == 1440
    classes = len({k for _, _, k in LAYER_WGRADS})
    mixed, panels, tiles = 0, 0, 0
    for x in range(8):
        seq = [(p, tm, tn) for q, p, tm, tn in plan if q == x]
        for i in range(0, len(seq), 32):
            wave = seq[i:i + 32]
            if len({LAYER_WGRADS[p][2] for p, _, _ in wave}) > 1:
                mixed += 1
                continue
            n = len({(p, "m", tm) for p, tm, _ in wave} | {(p, "n", tn) for p, _, tn in wave})
            if len(wave) == 32 and len({p for p, _, _ in wave}) == 1:
                assert n <= 13, (x, i, n)
            panels += n
            tiles += len(wave)
    assert mixed <= classes, mixed
    assert panels / tiles <= 0.42, panels / tiles   # (no sharing at all: 2.0; round 4's queues: 0.59 by the same count, before their start skew)


def test_grouped_weight_gradient_schedule_makespan(lib_path):
    """Greedy list scheduling of that plan on 32 workgroups per queue (cost of a tile = its K + a fixed 400-row overhead): the
    slowest workgroup is within 6 % of the ideal -- what replaces 7.9 rounds of split-K slabs plus the fold kernel."""
    import heapq
    from one_peace_amd import hip
    plan = hip.gemm_tn_grouped_plan(LAYER_WGRADS)
    worst, total = 0.0, 0.0
    for x in range(8):
        free = [0.0] * 32
        heapq.heapify(free)
        for q, p, _, _ in plan:
            if q == x:
                heapq.heappush(free, heapq.heappop(free) + LAYER_WGRADS[p][2] + 400.0)
        worst = max(worst, max(free))
        total += sum(LAYER_WGRADS[p][2] + 400.0 for q, p, _, _ in plan if q == x)
    assert total / 256.0 / worst > 0.94, (total / 256.0, worst)
