"""Build liblcs_hip.so (HIP kernels + C ABI) in-tree for gfx950.

    python -m lagrangiancoherence_amd.build

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the
GPU box with the working tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liblcs_hip.so")
SOURCES = ["api.hip", "pack.hip", "advect.hip", "sigma.hip", "ridges.hip", "halo.hip", "preprocess.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _strip_comments(src: str) -> str:
    """C/C++ source without comments and with runs of white space collapsed (string literals kept as they are)."""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == '"' or c == "'":                      # literal: copy through the closing quote
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1])
            i = j + 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def csrc_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources and the public header, comments and white space
    stripped: what a committed rocprof summary is stamped with, so a number replayed from profiles/ can be tied to
    the CODE it was measured on (editing a comment does not orphan the measurements)."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(HERE), "include", "lcs_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(_strip_comments(open(f, encoding="utf-8").read()).encode())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "lcs_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True, extra_flags=(), out: str | None = None) -> str:
    """``out``/``extra_flags``: experiment variants (e.g. ``-DLCS_LDS_READ2``) built beside the product
    library and selected at run time with ``LCS_LIB=<path>`` (see ``_capi.load``).

    Every translation unit is compiled to its own object (in parallel, cached under ``build/obj/<flags>/`` by the
    modification times of the source and the shared headers), then linked: editing one kernel file recompiles that
    file only."""
    if out is None and not force and not needs_build():
        return LIB
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    # the library carries the hash of the sources it was built from and the experiment flags it was built with
    # (lc_build_id()): a benchmark ties replayed counters to the BINARY that ran, not to the working tree
    build_id = csrc_hash() + ("" if not extra_flags else "+" + " ".join(sorted(extra_flags)))
    common = [_hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-fno-gpu-rdc", "-Wall",
              "-Wno-unused-function", "-Wno-pass-failed", *extra_flags]
    objdir = os.path.join(os.path.dirname(HERE), "build", "obj",
                          hashlib.sha256(" ".join(common).encode()).hexdigest()[:12])
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "lcs_hip.h"))
    hdr_t = max(os.path.getmtime(h) for h in headers)

    def compile_one(src):
        path, obj = os.path.join(CSRC, src), os.path.join(objdir, src + ".o")
        # api.hip holds lc_build_id(): its object depends on the id
        stamp = [f'-DLCS_BUILD_ID="{build_id}"'] if src == "api.hip" else []
        tag = obj + ".id"
        fresh = (not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t)
                 and (not stamp or (os.path.exists(tag) and open(tag).read() == build_id)))
        if fresh:
            return obj
        cmd = [*common, *stamp, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        if stamp:
            open(tag, "w").write(build_id)
        return obj
    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", out or LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out or LIB


if __name__ == "__main__":
    # python -m lagrangiancoherence_amd.build [--force] [--out path.so] [-Dflag ...]
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    flags = [a for a in argv if a.startswith("-D") or a.startswith("-m")]
    print(build_library(force="--force" in argv, extra_flags=flags, out=out))
