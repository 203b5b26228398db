"""Build libtce_hip.so (hand-written gfx950 kernels behind a C ABI) in-tree.

    python -m tce_rl_amd.build [--force]

Incremental by CONTENT HASH of (source, headers, flags, compiler), not by
modification time.

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


def _digest(paths, extra=()):
    """Content hash of the files (and flags) an object is built from."""
    import hashlib
    h = hashlib.sha256()
    for x in extra:
        h.update(str(x).encode() + b"\0")
    for p in sorted(paths):
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def build_library(force=False, verbose=True):
    """Compile what is stale, link.  Staleness is decided by CONTENT, not by
    modification time: every object is recorded (csrc/build/<name>.o.hash) with
    the hash of its source, all headers, its flags and the compiler path, so a
    fresh checkout next to shipped objects recompiles exactly when the sources
    differ from what the objects were built from -- and says which it did."""
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
               if f.endswith(".h")]
    api = os.path.join(os.path.dirname(PKG), "include", "tce_hip.h")
    if os.path.exists(api):                 # objective.hip checks itself against it
        headers.append(api)
    jobs, digests = [], {}
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        flags = FLAGS + FILE_FLAGS.get(src, [])
        d = digests[o] = _digest([s] + headers, [HIPCC] + flags)
        have = None
        try:
            with open(o + ".hash") as f:
                have = f.read().strip()
        except OSError:
            pass
        if force or not os.path.exists(o) or have != d:
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        with open(o + ".hash", "w") as f:
            f.write(digests[o] + "\n")
        if verbose:
            print("[tce_rl_amd.build] compiled", os.path.basename(s))

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, src[:-4] + ".o") for src in _sources()]
    link_d = _digest([], [digests[o] for o in objs])
    have = None
    try:
        with open(LIB + ".hash") as f:
            have = f.read().strip()
    except OSError:
        pass
    if force or jobs or not os.path.exists(LIB) or have != link_d:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] \
            + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr)
        with open(LIB + ".hash", "w") as f:
            f.write(link_d + "\n")
        if verbose:
            print("[tce_rl_amd.build] linked", LIB)
    elif verbose:
        print("[tce_rl_amd.build] up to date: %d objects match the content hash "
              "of their sources (%s)" % (len(objs), link_d))
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
