"""Build libtce_hip.so (hand-written gfx950 kernels behind a C ABI) in-tree.

    python -m tce_rl_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but
travels with the tree to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "csrc", "build")
LIB = os.path.join(PKG, "libtce_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17",
         "-Wno-unused-result"]
# per-file extras.  mlp16: the SLP vectorizer packs the split / activation
# arithmetic into v_pk_*_f32, which issue slower beside MFMAs (measured -5 %)
FILE_FLAGS = {"mlp16.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
               if f.endswith(".h")]
    api = os.path.join(os.path.dirname(PKG), "include", "tce_hip.h")
    if os.path.exists(api):                 # objective.hip checks itself against it
        headers.append(api)
    jobs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        if verbose:
            print("[tce_rl_amd.build] compiled", os.path.basename(s))

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, src[:-4] + ".o") for src in _sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] \
            + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
        if verbose:
            print("[tce_rl_amd.build] linked", LIB)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
