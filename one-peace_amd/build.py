"""Build libonepeace_hip.so (the C-ABI library) in-tree with hipcc for gfx950.

    python one-peace_amd/build.py [--force]

hipcc cross-compiles without a GPU, so this runs in the authoring container; the resulting .so is
git-ignored but travels to the GPU box with the repository snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libonepeace_hip.so")
PROBE_LIB = os.path.join(LIBDIR, "libonepeace_probe.so")  # hardware-semantics / power probes: test + measurement infrastructure,
PROBE_SRC = os.path.join("probes", "probe.hip")           # NOT part of the product library (include/onepeace_probe.h)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
         "-munsafe-fp-atomics"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [PROBE_SRC]:
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(LIBDIR, "build.sha256")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(PROBE_LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(cc, sources() + [PROBE_SRC]))
    probe_obj = objs.pop()
    common_obj = [o for o in objs if os.path.basename(o) == "capi_common.o"]  # the probe library carries its own error state
    for out, members in ((LIB, objs), (PROBE_LIB, [probe_obj] + common_obj)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + members
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stdout)
    open(stamp, "w").write(dig)
    if verbose:
        print("built", LIB, os.path.getsize(LIB) // 1024, "KiB")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
