"""The measurement scripts under tools/ and profiles/ only ever run on a GPU box; here: that they parse at all
(python: compiled, shell: `bash -n`), so that a typo does not cost a GPU call to find."""
import glob
import os
import py_compile
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = sorted(glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "profiles", "*.py")))
SH = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")) + glob.glob(os.path.join(ROOT, "profiles", "*.sh")))


@pytest.mark.parametrize("path", PY, ids=[os.path.relpath(p, ROOT) for p in PY])
def test_python_tool_compiles(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "out.pyc"), doraise=True)


@pytest.mark.parametrize("path", SH, ids=[os.path.relpath(p, ROOT) for p in SH])
def test_shell_tool_parses(path):
    r = subprocess.run(["bash", "-n", path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
