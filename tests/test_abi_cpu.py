"""The C-ABI library loads without a GPU and exports every symbol include/onepeace_hip.h declares; the ctypes table in
one-peace_amd/hip.py covers the same set with matching arity; the HIP path refuses to run without a device."""
import os
import re
import subprocess

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, "include", "onepeace_hip.h")).read()
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
    assert L.op_abi_version() == 1


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
