"""Build a variant of the product library from an alternative source of ONE csrc file (A/B timing of two code states on one GPU box):

    python tools/build_variant.py gemm.hip <path to the alternative gemm.hip> one-peace_amd/lib/libonepeace_hip_<tag>.so [-DFOO ...]

The other objects come from the regular build (one-peace_amd/build); select the variant at run time with ONEPEACE_HIP_LIB=<.so>.
"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("onepeace_build", os.path.join(ROOT, "one-peace_amd", "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def main():
    name, src, out = sys.argv[1:4]
    defs = sys.argv[4:]
    B.build(verbose=False)
    objdir = os.path.join(B.HERE, "build")
    obj = os.path.join(objdir, "variant_%s_%s.o" % (name[:-4], os.path.basename(out).replace(".so", "")))
    subprocess.run([B.HIPCC] + B.FLAGS + defs + ["-I", B.CSRC, "-c", "-x", "hip", src, "-o", obj], check=True)
    objs = [os.path.join(objdir, f[:-4] + ".o") for f in B.sources() if f != name] + [obj]
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
    print("built", out)


if __name__ == "__main__":
    main()
