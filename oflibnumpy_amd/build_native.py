"""Builds oflibnumpy_amd/libofl_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this also runs in the build container.  The .so is
git-ignored but travels to the GPU box with the repository snapshot.

Two libraries come out of the same sources:
  libofl_hip.so      the product: no environment knobs, no ablation probes, only the default kernels;
  libofl_hip_exp.so  the EXPERIMENTS build (-DOFL_EXPERIMENTS): the A/B kernel variants, ablation probes and OFL_* tuning /
                     test knobs.  Loaded only when OFL_LIB points at it (tools/, and the tests of the non-default routes).
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, os.environ.get("OFL_BUILD_NAME", "libofl_hip.so"))
OBJ_DIR = os.path.join(CSRC, "_obj" + os.environ.get("OFL_BUILD_TAG", ""))
EXP_OUT = os.path.join(HERE, "libofl_hip_exp.so")
EXP_OBJ_DIR = os.path.join(CSRC, "_obj_exp")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"] + os.environ.get("OFL_EXTRA_FLAGS", "").split()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X engine cannot be built on this machine")
    return exe


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_experiments(force=False, verbose=False):
    """The experiments build (see the module docstring).  Returns the library path."""
    return build_native(force, verbose, out=EXP_OUT, obj_dir=EXP_OBJ_DIR, extra=["-DOFL_EXPERIMENTS"])


def build_native(force=False, verbose=False, out=None, obj_dir=None, extra=()):
    """Compile every csrc/*.hip for gfx950 and link libofl_hip.so.  Returns the library path."""
    hipcc = _hipcc()
    OUT, OBJ_DIR, FLAGS = out or globals()["OUT"], obj_dir or globals()["OBJ_DIR"], globals()["FLAGS"] + list(extra)
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "ofl.h"))
    newest_hdr = max(os.path.getmtime(h) for h in headers)
    objs, relink = [], force or not os.path.exists(OUT)
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_hdr):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            relink = True
    if relink or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
    if "--experiments" in sys.argv:
        print(build_experiments(force="--force" in sys.argv, verbose=True))
