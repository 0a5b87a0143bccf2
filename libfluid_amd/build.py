"""Builds libfluid_amd/libfluid_amd.so (the C-ABI library, include/libfluid_amd.h) with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU box in-tree.
"""
import hashlib
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfluid_amd.so")
SOURCES = ["core.hip", "p2g.hip", "grid_ops.hip", "pcg.hip", "dist.hip", "particles.hip", "voxelizer.hip", "mesher.hip", "mg.hip", "pool.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=off",
         "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    """Content-driven: an object is recompiled when the hash of its source, of every header and of the compiler flags differs from
    the one recorded with it (csrc/.build_stamps.json) - file times play no part, so a checkout, a copy to another machine or a
    `touch` neither force nor hide a rebuild. Returns the library path; `last_build_report` says what was compiled and what was
    reused."""
    global last_build_report
    hipcc = _hipcc()
    # the compiler is part of an object's identity: after a ROCm upgrade (or another hipcc on the path) nothing is "reused"
    try:
        compiler = hipcc + "\n" + subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError:
        compiler = hipcc
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "libfluid_amd.h"))
    stamp_path = os.path.join(CSRC, ".build_stamps.json")
    try:
        with open(stamp_path) as f:
            stamps = json.load(f)
    except (OSError, ValueError):
        stamps = {}
    objs, jobs, todo = [], [], {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        d = _digest([s] + headers, " ".join(FLAGS) + "\n" + compiler)
        # (an object replaced by hand is not trusted either: its own hash is kept beside its source digest)
        if force or not os.path.exists(o) or stamps.get(src) != d or stamps.get(src + ":obj") != _digest([o]):
            jobs.append([hipcc, *FLAGS, "-c", s, "-o", o])
            todo[src] = d

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for w in ex.map(run, jobs):
            if verbose and w:
                print(w)
    link_digest = _digest(objs)
    relinked = force or bool(jobs) or not os.path.exists(LIB) or stamps.get("__link__") != link_digest
    if relinked:
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl", "-lpthread"])
    stamps.update(todo)
    for src in todo:
        stamps[src + ":obj"] = _digest([os.path.join(CSRC, src.replace(".hip", ".o"))])
    stamps["__link__"] = link_digest
    with open(stamp_path, "w") as f:
        json.dump(stamps, f, indent=1, sort_keys=True)
    last_build_report = {"compiled": sorted(todo), "reused": sorted(set(SOURCES) - set(todo)), "relinked": relinked}
    return LIB


last_build_report = None


def build_variant(name, defs, sources):
    """A/B builds: `sources` recompiled with extra -D switches, linked with the other objects of the regular build into
    libfluid_amd/variants/<name>.so (select it with LFA_LIB_PATH; the regular library is never overwritten)."""
    build()
    hipcc = _hipcc()
    vdir = os.path.join(HERE, "variants")
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if src in sources:
            o = os.path.join(vdir, name + "_" + src.replace(".hip", ".o"))
            r = subprocess.run([hipcc, *FLAGS, *["-D" + d for d in defs], "-c", os.path.join(CSRC, src), "-o", o],
                               capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        objs.append(o)
    lib = os.path.join(vdir, name + ".so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs, "-ldl", "-lpthread"], check=True)
    return lib


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "variant":  # python build.py variant NAME p2g.hip[,core.hip] -DX=1 ...
        print(build_variant(sys.argv[2], [a[2:] for a in sys.argv[4:]], sys.argv[3].split(",")))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
        print(last_build_report)
