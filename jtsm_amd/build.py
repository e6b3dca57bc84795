"""Build libjtsm_hip.so (gfx950) from jtsm_amd/csrc/*.hip with hipcc.

    python -m jtsm_amd.build            # incremental: objects are rebuilt when sources change
    python -m jtsm_amd.build --force

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
jtsm_amd/lib/libjtsm_hip.so is git-ignored but travels to the GPU box with the tree.
"""
import argparse
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libjtsm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"] + os.environ.get("JTSM_EXTRA_HIPCC_FLAGS", "").split()
# Per-file flags.  semseg_ops.hip without the SLP vectoriser, i.e. without packed fp32 arithmetic (v_pk_fma_f32 ...):
# with the semantic head on its own stream (modeling/meta_arch/mcnn.py) the bilinear up-sampling kernel built WITH it
# stored sums that lacked one of their four terms in lanes 48-63 of about 1 % of its wavefronts whenever a second
# hardware queue was busy ("the side-stream anomaly": 10-11 of 12 forward passes wrong; 0 of 35 without
# packed arithmetic; idle cycles after the loads / before the store change nothing).  These kernels are bound by
# their bytes: the scalar form costs nothing measurable.
FILE_FLAGS = {"semseg_ops.hip": ["-fno-slp-vectorize"]}


def _stale(target, deps):
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, jobs=4):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
        os.path.join(HERE, "..", "include", "jtsm_hip.h")]
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s, os.path.abspath(__file__)] + hdrs):
            todo.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        return s, r.returncode, r.stdout

    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            for s, rc, out in ex.map(cc, todo):
                if out.strip() and (verbose or rc):
                    print(out, file=sys.stderr)
                if rc:
                    raise RuntimeError("hipcc failed on %s" % s)
    if todo or force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, verbose=a.verbose))
