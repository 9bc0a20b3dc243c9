"""Static check of the gfx950 ISA hipcc produced for csrc/gemm.hip (CPU only, needs hipcc):

    python tools/check_mfma_hazards.py [path/to/gemm-hip-amdgcn-amd-amdhsa-gfx950.s]

The four-wave GEMM kernels issue their MFMAs as inline asm (accumulators pinned in AGPRs).  hipcc treats an asm statement as
one opaque instruction: its hazard recogniser does NOT put the wait states between such an MFMA and a later instruction that
reads or overwrites its accumulator (an 8-pass MFMA needs 18 before a v_accvgpr_read / v_accvgpr_mov of its result).  Inside
straight-line code the kernels pad with s_nop themselves; what cannot be controlled from the source are the copies the register
allocator puts on control-flow edges (a peeled loop tail made it shuffle accumulators right behind the loop's last MFMAs ->
a few accumulator registers per wave were read stale, deterministically per binary; found in round 3).  This script walks
every kernel's instruction stream in program order (every loop body a second time, for hazards across its back edge) and
reports any compiler-generated instruction that touches an AGPR fewer than NEED wait states after the inline-asm MFMA that
writes that AGPR.  Without an argument it compiles csrc/gemm.hip with -save-temps in a temporary directory first.
Exit status 0 = clean.
"""
import os
import re
import subprocess
import sys
import tempfile

NEED = 18  # wait states between an 8-pass XDL write and a non-MFMA read / overwrite of the result
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AGPR = re.compile(r"\ba(?:\[(\d+):(\d+)\]|(\d+)\b)")


def agprs(text):
    out = []
    for m in AGPR.finditer(text):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def compile_isa(workdir):
    src = os.path.join(ROOT, "one-peace_amd", "csrc", "gemm.hip")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
           "-munsafe-fp-atomics", "-save-temps", "-c", src, "-o", os.path.join(workdir, "gemm.o")]
    subprocess.run(cmd, cwd=workdir, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return os.path.join(workdir, "gemm-hip-amdgcn-amd-amdhsa-gfx950.s")


def check_lines(lines):
    """lines: the assembly text.  Returns (number of functions, [problem strings])."""
    problems, kernels = [], 0
    name, in_asm, now, wrote, labels, redone = None, False, 0, {}, {}, set()
    i = 0
    while i < len(lines):
        raw = lines[i]
        line = raw.strip()
        i += 1
        m = re.match(r"^(_Z\w+):", raw)
        if m:  # a function label
            name, in_asm, now, wrote, labels, redone = m.group(1), False, 0, {}, {}, set()
            kernels += 1
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            labels[m.group(1)] = i  # index of the first line after the label
            continue
        if not line or line.startswith("."):
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        if line.startswith(";"):
            continue
        code = line.split(";")[0]
        op = code.split()[0]
        if in_asm:
            if op.startswith("v_mfma"):
                for r in agprs(code.split(None, 1)[1].split(",")[0]):
                    wrote[r] = now
        else:
            for r in agprs(code):
                if r in wrote and now - wrote[r] < NEED:
                    problems.append("%s: line %d: `%s` touches a%d %d wait states after the inline-asm MFMA that writes it" % (
                        name, i, code.strip(), r, now - wrote[r]))
                    break
            if op.startswith("s_cbranch") or op == "s_branch":
                target = code.split()[-1]
                if target in labels and (i, target) not in redone:  # backward branch: walk the loop body once more
                    redone.add((i, target))
                    i = labels[target]
                    continue
        now += (int(code.split()[1]) + 1) if op == "s_nop" else 1
    return kernels, problems


def check(path):
    return check_lines(open(path).read().split("\n"))


def main():
    if len(sys.argv) > 1:
        kernels, problems = check(sys.argv[1])
    else:
        with tempfile.TemporaryDirectory() as d:
            kernels, problems = check(compile_isa(d))
    per_kernel = {}
    for p in problems:
        per_kernel.setdefault(p.split(":")[0], []).append(p)
    for k, ps in per_kernel.items():
        print("%s: %d hazards, first: %s" % (k, len(ps), ps[0].split(": ", 1)[1]))
    print("%d functions scanned, %d hazards" % (kernels, len(problems)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
