"""CPU suite: the C-ABI library loads here (no GPU) and exports every symbol the header declares;
the product never reaches into oracle/."""
import os
import re

import pytest

from conftest import ROOT


def _declared():
    names = set()
    inc = os.path.join(ROOT, "include")
    for f in os.listdir(inc):
        if f.endswith(".h"):
            txt = open(os.path.join(inc, f)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(jtsm_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_loads_and_exports_all_declared_symbols():
    from jtsm_amd import _lib

    lib = _lib.lib()
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    assert len(_declared()) >= 15
    assert b"gfx950" in lib.jtsm_version()


def test_code_object_targets_gfx950_only():
    from jtsm_amd import _lib

    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_80"):
        assert other not in blob


def test_cpu_tensors_are_refused_not_silently_computed():
    import torch

    from jtsm_amd.layers import MOIPool, ROIAlign, ROIAlignRotated

    with pytest.raises(RuntimeError):
        ROIAlign((7, 7), 1.0, 0)(torch.zeros(1, 1, 5, 5), torch.zeros(1, 5))
    with pytest.raises(RuntimeError):
        ROIAlignRotated((7, 7), 1.0, 0)(torch.zeros(1, 1, 5, 5), torch.zeros(1, 6))
    with pytest.raises(RuntimeError):
        MOIPool((7, 7), 1.0)(torch.zeros(1, 1, 5, 5), torch.zeros(1, 5),
                             torch.zeros(1, 4, dtype=torch.int32),
                             torch.zeros(1, 20, 20, dtype=torch.int32))


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "jtsm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "liboracle" in txt \
                        or re.search(r'#include\s+".*oracle', txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
