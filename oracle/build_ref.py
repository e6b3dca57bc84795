"""ORACLE — TEST INFRASTRUCTURE ONLY.

Builds oracle/_ref/jtsm_ref_cpu.so from the reference's own CPU sources, compiled where
they lie under /root/reference (never copied), plus oracle/ref_shim.cpp (ours).  Build-container
only: the module pins the C restatement (tests/test_oracle_pooling.py) and generates the committed
vectors (tests/golden/make_golden.py); those .npz files are the pin on the GPU box.  The built
files are TRANSIENT — __graft_entry__.build() and the test session remove them again (clean()),
so no compiled reference code travels to the GPU box, and load_prebuilt() refuses to load one
where /root/reference is absent.

    python oracle/build_ref.py        # or: make -C oracle ref
"""
import os
import sys

REF = os.environ.get("JTSM_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
NAME = "jtsm_ref_cpu"


def build(verbose=False):
    csrc = os.path.join(REF, "detectron2", "layers", "csrc")
    srcs = [
        os.path.join(csrc, "ROIAlign", "ROIAlign_cpu.cpp"),
        os.path.join(csrc, "ROIAlignRotated", "ROIAlignRotated_cpu.cpp"),
        os.path.join(csrc, "nms_rotated", "nms_rotated_cpu.cpp"),   # the reference's greedy NMS loop (angle 0 = plain boxes)
    ]
    if not all(os.path.isfile(s) for s in srcs):
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load

    # -ffp-contract=off: plain x86-64 has no FMA, so this only documents the intent.
    return load(
        name=NAME,
        sources=srcs + [os.path.join(HERE, "ref_shim.cpp")],
        extra_include_paths=[csrc],
        extra_cflags=["-O2", "-ffp-contract=off"],
        build_directory=OUT,
        verbose=verbose,
    )


def clean():
    """Remove every built file under oracle/_ref (the .so, its objects, the ninja files)."""
    import shutil

    shutil.rmtree(OUT, ignore_errors=True)


def load_prebuilt():
    """Import the already-built module from oracle/_ref — in the build container only."""
    import glob
    import importlib.util

    if not os.path.isdir(REF):
        return None

    import torch  # noqa: F401  (libtorch symbols must be loaded first)

    hits = sorted(glob.glob(os.path.join(OUT, NAME + "*.so")))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location(NAME, hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build(verbose=True)
    if m is None:
        print("reference sources not found under", REF, file=sys.stderr)
        sys.exit(1)
    print("built", m.__file__ if hasattr(m, "__file__") else m)
