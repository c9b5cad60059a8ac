"""Build libc2s_hip.so (gfx950) in-tree with hipcc.  No torch dependency: the library is a plain C ABI."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libc2s_hip.so")
SOURCES = ["conv_igemm.hip", "conv_bf16x3.hip", "conv_xpair.hip", "conv_first.hip", "conv_winograd.hip", "conv_winograd16.hip", "conv_s2wino.hip", "conv_s2dgrad.hip", "conv_wgrad.hip", "norm.hip", "se.hip", "ltae.hip", "ltae_pe.hip", "aggregate.hip", "misc.hip", "metrics.hip", "io.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-unused-value", "-D__HIP_PLATFORM_AMD__"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    cc = hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(os.path.dirname(HERE), "include", "c2s_hip.h")]
    objs, jobs = [], []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [cc, *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if force or jobs or _stale(LIB, objs):
        r = subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
