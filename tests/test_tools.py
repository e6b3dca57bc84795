"""CPU suite: the measurement tools and driver hooks at least parse and expose their command lines (their runs
need the GPU box)."""
import ast
import glob
import os
import subprocess
import sys

from conftest import ROOT


def test_tool_sources_parse():
    files = glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "tools", "sweeps", "*.py")) + \
        [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]
    assert len(files) >= 10
    for f in files:
        ast.parse(open(f).read(), filename=f)


def test_bench_command_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def test_make_profiles_script_is_valid_shell():
    r = subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", "make_profiles.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
