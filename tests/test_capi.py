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
    # offload-bundle entries name their target; rocPRIM's host-side dispatch tables carry other architectures' NAMES
    # as plain strings, which is not code for them
    for other in (b"gfx942", b"gfx90a", b"gfx1100"):
        assert b"amdgcn-amd-amdhsa--" + other not in blob
    assert b"sm_80" not in blob and b"nvptx" not in blob


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


def test_conv_planning_host_logic_handles_empty_and_odd_shapes():
    """jtsm_conv_workspace_bytes / jtsm_conv_plan are pure host code: an empty batch (a rank with no
    foreground roi) must plan to 'nothing', not divide by zero; a real layer plans a sane tile / split."""
    import ctypes as C
    from jtsm_amd import _lib
    from jtsm_amd.layers.conv import ConvShape

    lib = _lib.lib()
    lib.jtsm_conv_workspace_bytes.restype = C.c_size_t
    for kh, stride, pad in ((1, 1, 0), (3, 1, 1), (1, 2, 0), (2, 2, 0)):
        s = ConvShape(0, 14, 14, 256, 256, kh, kh, stride, pad, 1)
        for bwd in (0, 1):
            assert lib.jtsm_conv_workspace_bytes(C.byref(s), bwd) == 0
        for role in (0, 1, 2):
            k, tm, tn, sp = C.c_int(-1), C.c_int(), C.c_int(), C.c_int()
            assert lib.jtsm_conv_plan(C.byref(s), role, 0, C.byref(k), C.byref(tm), C.byref(tn), C.byref(sp)) == 0
            assert sp.value == 1
    # res5 3x3 at 2x(32x32): few output tiles -> split-K with a workspace of splits * M * N floats
    s = ConvShape(2, 32, 32, 512, 512, 3, 3, 1, 1, 1)
    k, tm, tn, sp = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert lib.jtsm_conv_plan(C.byref(s), 0, 0, C.byref(k), C.byref(tm), C.byref(tn), C.byref(sp)) == 0
    assert (tm.value, tn.value) == (128, 128) and 1 < sp.value <= 16
    assert lib.jtsm_conv_workspace_bytes(C.byref(s), 0) == sp.value * 2048 * 512 * 4
    bad = ConvShape(1, 8, 8, 3, 8, 1, 1, 1, 0, 1)  # in_c not a multiple of 4
    assert lib.jtsm_conv_plan(C.byref(bad), 0, 0, None, None, None, None) != 0
    assert b"in_c" in lib.jtsm_last_error()


def test_bf16x3_plan_reports_the_ring_for_long_k_64_tiles():
    """jtsm_conv_bf16x3_plan (host code): the res4 / res5 1x1 layers that run on 64 x 64 tiles take the four-stage
    ring (NBUF = 4) from four stages per K slice; the GPU conv cases `ring_*` in tests/test_hip_conv.py are such
    shapes; large layers stay double-buffered on 256 x 256 tiles."""
    import ctypes as C
    from jtsm_amd import _lib
    from jtsm_amd.layers.conv import ConvShape

    lib = _lib.lib()

    def plan(shape, role):
        v = [C.c_int() for _ in range(6)]
        assert lib.jtsm_conv_bf16x3_plan(C.byref(ConvShape(*shape)), role, *[C.byref(x) for x in v]) == 0
        return tuple(x.value for x in v)   # wm, wn, tm, tn, nbuf, splits

    # BASELINE layers: res4 conv1 (1024 -> 256 on 2 x 64 x 64), res5 conv1 (2048 -> 512 on 2 x 32 x 32), forward
    assert plan((2, 64, 64, 1024, 256, 1, 1, 1, 0, 1), 0) == (2, 2, 1, 1, 4, 1)
    assert plan((2, 32, 32, 2048, 512, 1, 1, 1, 0, 1), 0) == (2, 2, 1, 1, 4, 2)
    # data gradient of res4 conv3 (256 -> 1024): contracted over the 1024 output channels
    assert plan((2, 64, 64, 256, 1024, 1, 1, 1, 0, 1), 1) == (2, 2, 1, 1, 4, 1)
    # the GPU suite's ring cases
    assert plan((2, 16, 16, 1024, 256, 1, 1, 1, 0, 1), 0) == (2, 2, 1, 1, 4, 8)
    assert plan((1, 20, 20, 352, 128, 1, 1, 1, 0, 1), 0) == (2, 2, 1, 1, 4, 2)
    assert plan((1, 12, 12, 64, 128, 3, 3, 1, 1, 1), 0) == (2, 2, 1, 1, 4, 4)
    # FPN p2 output conv: the halo kernel (reported as NBUF = 0); a res2-sized 1x1 (8 stages, 2048 tiles): 128 x 128
    # tiles, double-buffered
    assert plan((2, 256, 256, 256, 256, 3, 3, 1, 1, 1), 0)[4] == 0
    assert plan((2, 256, 256, 256, 256, 1, 1, 1, 0, 1), 0)[:5] == (2, 2, 2, 2, 2)


def test_integration_doc_maps_every_declared_symbol():
    """INTEGRATION.md's symbol <-> reference-interface table names every entry point include/jtsm_hip.h declares
    (brace lists and trailing-* families expanded)."""
    import re

    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    have = set()
    for m in re.findall(r"jtsm_[a-z0-9_{},*]+", integ):
        parts = [m]
        while any("{" in p for p in parts):
            nxt = []
            for p in parts:
                mm = re.search(r"\{([^{}]*)\}", p)
                if not mm:
                    nxt.append(p)
                    continue
                nxt += [p[:mm.start()] + alt + p[mm.end():] for alt in mm.group(1).split(",")]
            parts = nxt
        have.update(parts)
    missing = [n for n in _declared()
               if n not in have and not any("*" in w and re.fullmatch(w.replace("*", ".*"), n) for w in have)]
    assert not missing, missing
