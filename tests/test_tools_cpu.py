"""The measurement scripts under tools/ only run on the GPU box; here they must at least compile, and the shell scripts must
reference files that exist (a renamed kernel flavour or tool otherwise shows up as a wasted GPU call)."""
import glob
import os
import py_compile
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tool_scripts_compile():
    files = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")))
    assert len(files) > 20
    for f in files:
        py_compile.compile(f, doraise=True)


def test_shell_scripts_reference_existing_tools():
    for sh in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))):
        text = open(sh).read()
        for rel in re.findall(r"tools/([A-Za-z0-9_]+\.(?:py|sh))", text):
            assert os.path.exists(os.path.join(ROOT, "tools", rel)), (os.path.basename(sh), rel)
