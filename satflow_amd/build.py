"""Build libsatflow_hip.so (gfx950) in-tree with hipcc.  ``python -m satflow_amd.build [--force]``."""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import re
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libsatflow_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG, "..", "include", "*.h"))
    jobs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        # a translation unit that #includes another .hip (the fp16 builds of the bf16 kernels) is stale when that file changes
        included = [os.path.join(CSRC, m) for m in re.findall(r'#include "([^"]+\.hip)"', open(src).read())]
        if force or _stale(obj, [src] + included + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    with cf.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(compile_one, jobs):
            if verbose:
                print(f"[hipcc] {os.path.basename(src)} rc={rc}")
            if rc != 0:
                raise RuntimeError(f"hipcc failed for {src}:\n{out}")
            if out.strip() and verbose:
                print(out)
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in _sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}{r.stderr}")
        if verbose:
            print(f"[link] {LIB}")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
