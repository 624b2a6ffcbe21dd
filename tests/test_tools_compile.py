"""The profiling / A-B drivers under tools/ and the two root scripts at least parse: they are run by hand on the GPU box,
so nothing else would notice one that rotted with a refactor."""
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPTS = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]


@pytest.mark.parametrize("path", SCRIPTS, ids=[os.path.basename(p) for p in SCRIPTS])
def test_script_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "out.pyc"), doraise=True)


def test_shell_drivers_parse():
    import subprocess

    for path in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))):
        subprocess.run(["bash", "-n", path], check=True)
