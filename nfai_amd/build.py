"""Build libnfai_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m nfai_amd.build [--force] [--jobs N]

One object per .hip translation unit (parallel), then one shared link.  Objects are rebuilt only
when their source or a header is newer.  The .so is git-ignored but travels with `gpurun`.
"""
from __future__ import annotations

import argparse
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
OUT = os.path.join(CSRC, "libnfai_hip.so")
OBJ_DIR = os.path.join(CSRC, "build")
ARCH = "gfx950"

SOURCES = ["api.hip", "kernels_basic.hip", "kernels_gemv.hip", "kernels_gemv_kq.hip", "kernels_gemv_kqm.hip", "kernels_attn.hip", "kernels_engine.hip", "kernels_prefill.hip", "llama.hip", "pp.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "kqm.h"), os.path.join(ROOT, "include", "nfai_hip.h")]

CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fvisibility=hidden", "-Wall",
            "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
            "-ffp-contract=off",  # a*b+c stays two roundings unless written as fmaf (parity bookkeeping)
            "-fno-fast-math"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP backend cannot be built (there is no CPU fallback)")


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, jobs: int | None = None, verbose: bool = False, extra_sources: list[str] | None = None,
          stamps: bool = False) -> str:
    """stamps=True builds the DIAGNOSTIC library libnfai_hip_stamps.so (-DNFAI_STAMPS: s_memrealtime readings inside the decode
    kernels, tools/stamps.py); the product library never contains a stamp."""
    global OUT, OBJ_DIR
    if stamps:
        out, obj_dir, flags = os.path.join(CSRC, "libnfai_hip_stamps.so"), os.path.join(CSRC, "build", "stamps"), ["-DNFAI_STAMPS"]
        saved = (OUT, OBJ_DIR, list(CXXFLAGS))
        OUT, OBJ_DIR = out, obj_dir
        CXXFLAGS.extend(flags)
        try:
            return build(force, jobs, verbose, extra_sources)
        finally:
            OUT, OBJ_DIR = saved[0], saved[1]
            CXXFLAGS[:] = saved[2]
    os.makedirs(OBJ_DIR, exist_ok=True)
    cc = hipcc()
    sources = SOURCES + [s for s in (extra_sources or []) if s not in SOURCES]
    sources = [s for s in sources if os.path.exists(os.path.join(CSRC, s))]
    todo = []
    for s in sources:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + HEADERS):
            todo.append((src, obj))

    def compile_one(pair):
        src, obj = pair
        cmd = [cc] + CXXFLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(4, len(todo))) as ex:
            list(ex.map(compile_one, todo))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in sources]
    if force or todo or _stale(OUT, objs):
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("-v", "--verbose", action="store_true")
    ap.add_argument("--stamps", action="store_true", help="build the diagnostic libnfai_hip_stamps.so instead")
    a = ap.parse_args()
    print(build(a.force, a.jobs, a.verbose, stamps=a.stamps))
