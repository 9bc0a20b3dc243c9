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

Second check (round 3, csrc/gemm.hip epilogue_v): `buffer_store_dwordx3/x4 v[a:b], v, s[..], sN offen` reads its data VGPRs
late; this hipcc does not keep a following VALU write of v[a:b] away from it when the offset is an SGPR (dword 1 of lanes 12-15
of each 16-lane row went out overwritten).  Reported: any instruction that writes one of the data VGPRs fewer than STORE_NEED
wait states behind such a store, in fall-through order (labels are walked through: the miss was first seen across one).
Third check: those stores are inline asm, so the compiler does not give them the SGPR_NEED wait states a VMEM instruction needs
behind an SALU / v_readfirstlane write of an SGPR it reads (row offset, descriptor) -- a store went out with the previous row's
offset; reported when the asm block does not supply them itself.
Exit status 0 = clean.
"""
import os
import re
import subprocess
import sys
import tempfile

NEED = 18  # wait states between an 8-pass XDL write and a non-MFMA read / overwrite of the result
NEED_F8 = 34  # ... a 16-pass one (v_mfma_scale_f32_16x16x128_f8f6f4 on fp8 operands: twice as long in the pipe)
SGPR_NEED = 5   # wait states between an SALU / v_readfirstlane write of an SGPR and an inline-asm VMEM instruction that reads it
STORE_NEED = 2  # wait states between a >= 12-byte buffer store with an SGPR offset and a VALU write of its data VGPRs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AGPR = re.compile(r"\ba(?:\[(\d+):(\d+)\]|(\d+)\b)")
VGPR = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+)\b)")
WIDE_STORE = re.compile(r"^buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*v\d+,\s*s\[\d+:\d+\],\s*s\d+")


def vgprs(text):
    out = []
    for m in VGPR.finditer(text):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_store_data(lines):
    """The store-data rule over the whole file, fall-through order.  Returns [problem strings]."""
    problems, name = [], None
    pending = []  # (set of data VGPRs, wait states since the store, text)
    for n, raw in enumerate(lines, 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            name, pending = m.group(1), []
            continue
        if not line or line.startswith((".", ";")) or line.endswith(":"):
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        if op == "s_nop":
            step = int(code.split()[1]) + 1
        else:
            step = 1
            if op.startswith("v_") and not op.startswith("v_cmp") and len(code.split(None, 1)) > 1:
                dest = set(vgprs(code.split(None, 1)[1].split(",")[0]))
                for regs, since, text in pending:
                    if since < STORE_NEED and dest & regs:
                        problems.append("%s: line %d: `%s` overwrites data of `%s` %d wait states behind it" % (name, n, code, text, since))
        pending = [(r, since + step, t) for r, since, t in pending if since + step < STORE_NEED]
        m = WIDE_STORE.match(code)
        if m:
            pending.append((set(range(int(m.group(1)), int(m.group(2)) + 1)), 0, code))
    return problems



def agprs(text):
    out = []
    for m in AGPR.finditer(text):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def compile_isa(workdir, name="gemm"):
    src = os.path.join(ROOT, "one-peace_amd", "csrc", name + ".hip")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
           "-munsafe-fp-atomics", "-save-temps", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", os.path.join(workdir, name + ".o")]
    r = subprocess.run(cmd, cwd=workdir, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    open(os.path.join(workdir, name + ".resource_usage.txt"), "w").write(r.stdout)  # the remarks of the same compile (resource_usage)
    return os.path.join(workdir, name + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def resource_usage(isa_path):
    """{mangled kernel name: {"VGPRs", "AGPRs", "SGPRs", "ScratchSize", "Occupancy", "LDS"}} from the kernel-resource-usage remarks
    compile_isa stored next to the assembly."""
    txt = open(isa_path.replace("-hip-amdgcn-amd-amdhsa-gfx950.s", ".resource_usage.txt")).read()
    out, cur = {}, None
    keys = {"VGPRs": "VGPRs", "AGPRs": "AGPRs", "SGPRs": "SGPRs", "ScratchSize": "ScratchSize", "Occupancy": "Occupancy", "LDS Size": "LDS"}
    for line in txt.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for k, short in keys.items():
            m = re.search(r"remark: .*?" + k + r"[^:]*: (\d+)", line)
            if m and short not in cur:
                cur[short] = int(m.group(1))
    return out


VM_OP = re.compile(r"^(global_|buffer_|flat_|scratch_)(load|store|atomic)")
WAIT_VM = re.compile(r"s_waitcnt\b.*\bvmcnt\((\d+)\)")


def check_inflight_asm_loads(lines):
    """Round 4 (csrc/attention.hip, persistent kernels): loads issued as inline asm write their destination registers some hundred
    cycles AFTER the statement, and the compiler -- for which the statement's outputs are defined at once -- is free to copy or
    re-use those registers.  The kernels tie them to their hand-placed `s_waitcnt vmcnt(N)` statements; this walks every function
    in fall-through order, counts vector-memory operations (loads, stores and LDS-DMA complete in issue order), and reports any
    instruction outside an asm block that reads or writes a register an inline-asm load has not provably delivered yet (a wait
    vmcnt(N) delivers everything but the N youngest operations).  A linear walk, not a control-flow analysis: behind an unconditional
    branch nothing is assumed to be in flight, so it can miss a hazard on a jumped-to path but does not invent one; what it is for --
    copies the register allocator puts at a loop header or in a fall-through block -- it sees.  Returns [problem strings]."""
    problems, name, in_asm, issued, inflight = [], None, False, 0, {}
    for n, raw in enumerate(lines, 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            name, in_asm, issued, inflight = m.group(1), False, 0, {}
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        if not line or line.startswith((".", ";")) or line.endswith(":"):
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        if op in ("s_branch", "s_endpgm", "s_setpc_b64"):  # what follows is reached from elsewhere: nothing is known there (lenient)
            inflight = {}
            continue
        w = WAIT_VM.search(code)
        if w:
            done = issued - int(w.group(1))
            inflight = {r: i for r, i in inflight.items() if i > done}
            continue
        if op == "s_waitcnt" and "vmcnt" not in code and not in_asm:
            continue
        regs = vgprs(code.split(None, 1)[1]) if len(code.split(None, 1)) > 1 else []
        if not (in_asm and op.startswith("global_load")):
            hit = [r for r in regs if r in inflight]
            if hit and not in_asm:
                problems.append("%s: line %d: `%s` touches v%d while the inline-asm load of it is in flight" % (name, n, code, hit[0]))
                for r in hit:
                    inflight.pop(r, None)
        if VM_OP.match(op):
            issued += 1
            if in_asm and op.startswith("global_load") and len(code.split(None, 1)) > 1:
                for r in vgprs(code.split(None, 1)[1].split(",")[0]):
                    inflight[r] = issued
    return problems


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
                # (the fp8 MX form v_mfma_scale_f32_16x16x128_f8f6f4 runs 8 passes, twice the 16x16x32 bf16 form: its result is
                # NEED_F8 wait states away; `wrote` keeps the time from which the register may be touched)
                for r in agprs(code.split(None, 1)[1].split(",")[0]):
                    wrote[r] = now + (NEED_F8 - NEED if "f8f6f4" in op else 0)
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


SGPR = re.compile(r"\bs(?:\[(\d+):(\d+)\]|(\d+)\b)")


def sgprs(text):
    out = []
    for m in SGPR.finditer(text):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_asm_vmem_sgprs(lines):
    """SGPR write -> inline-asm VMEM read, fall-through order.  Returns [problem strings]."""
    problems, name, in_asm, now, wrote = [], None, False, 0, {}
    for n, raw in enumerate(lines, 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            name, in_asm, now, wrote = m.group(1), False, 0, {}
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        if not line or line.startswith((".", ";")) or line.endswith(":"):
            continue
        code = line.split(";")[0].strip()
        if not code:
            continue
        op = code.split()[0]
        if in_asm and op.startswith(("buffer_", "global_", "flat_")):
            for r in sgprs(code):
                if r in wrote and now - wrote[r] < SGPR_NEED:
                    problems.append("%s: line %d: inline-asm `%s` reads s%d %d wait states after it was written" % (name, n, code, r, now - wrote[r]))
                    break
        elif (op.startswith("s_") and not op.startswith(("s_nop", "s_waitcnt", "s_cmp", "s_cbranch", "s_branch", "s_barrier", "s_endpgm", "s_load", "s_sleep", "s_setprio"))) \
                or op == "v_readfirstlane_b32" or op == "v_readlane_b32":
            rest = code.split(None, 1)
            if len(rest) > 1:
                for r in sgprs(rest[1].split(",")[0]):
                    wrote[r] = now
        now += (int(code.split()[1]) + 1) if op == "s_nop" else 1
    return problems


def check(path):
    lines = open(path).read().split("\n")
    kernels, problems = check_lines(lines)
    return kernels, problems + check_store_data(lines) + check_asm_vmem_sgprs(lines) + check_inflight_asm_loads(lines)


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
