"""Static checks on the gfx950 ISA hipcc produces for the GEMM kernels (no GPU needed; compiles csrc/gemm.hip once)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools import check_mfma_hazards as C  # noqa: E402


def test_hazard_checker_sees_a_stale_accumulator_read():
    bad = """_Z3foov:
\t;;#ASMSTART
\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]
\t;;#ASMEND
\ts_nop 3
\tv_accvgpr_read_b32 v9, a2
\ts_endpgm"""
    assert len(C.check_lines(bad.split("\n"))[1]) == 1
    assert C.check_lines(bad.replace("s_nop 3", "s_nop 7\n\ts_nop 7\n\ts_nop 7").split("\n"))[1] == []
    loop = """_Z3barv:
.LBB0_1:
\tv_accvgpr_mov_b32 a8, a1
\ts_nop 7
\ts_nop 7
\ts_nop 7
\t;;#ASMSTART
\tv_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm"""
    assert len(C.check_lines(loop.split("\n"))[1]) == 1  # only visible across the back edge


@pytest.mark.skipif(shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None, reason="needs hipcc")
def test_no_compiler_instruction_touches_an_accumulator_behind_an_inline_asm_mfma(tmp_path):
    """The four-wave kernels pin their accumulators in AGPRs through inline-asm MFMAs the compiler's hazard recogniser cannot
    see; copies it places on control-flow edges must come at least 18 wait states after the MFMA that writes the register."""
    kernels, problems = C.check(C.compile_isa(str(tmp_path)))
    assert kernels > 20
    assert problems == [], problems[:5]
