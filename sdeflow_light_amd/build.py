"""Build libmsgm_hip.so (gfx950) from csrc/*.hip with hipcc, in-tree.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmsgm_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-ffp-contract=off"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; libmsgm_hip.so cannot be built")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "msgm_hip.h")]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        if force or _stale(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s}:\n{r.stderr}")
        return s

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for s in ex.map(cc, jobs):
                if verbose:
                    print(f"[build] compiled {os.path.basename(s)}", file=sys.stderr)
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
