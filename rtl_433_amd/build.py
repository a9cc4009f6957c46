"""Builds librtl433hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m rtl_433_amd.build [--force]

hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INC = os.path.join(HERE, "..", "include")
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "librtl433hip.so")
SEAM_OUT = os.path.join(OUT_DIR, "librtl433seam.so")  # the reference's own function names over the C ABI (csrc/ref_seam.cpp)

SOURCES = ["stream_kernels.hip", "slicer_kernels.hip", "baseband_kernels.hip", "analyzer_kernels.hip", "host_api.cpp", "batch_run.cpp",
           "dispatch.cpp", "reports.cpp", "pulse_text.cpp", "prefilter.cpp", "filter_frame.cpp", "detect_seam.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function", "-x", "hip"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale():
    if not os.path.exists(OUT) or not os.path.exists(SEAM_OUT):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(SEAM_OUT))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INC, f) for f in os.listdir(INC)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, timing=False):
    """timing=True: the development variant lib/librtl433hip_timing.so whose detection kernel keeps per-phase shader
    clocks (R433_DEBUG_TIMING, tools/kbench.py --debug 1024); the product library is built without that code."""
    if timing:
        return _build_variant("librtl433hip_timing.so", ["-DR433_KERNEL_TIMING"], verbose)
    if not force and not _stale():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(OUT_DIR, src.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc] + FLAGS + ["-I", INC, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", OUT] + objs
    subprocess.check_call(cmd)
    for o in objs:
        os.remove(o)
    build_seam(OUT_DIR, "rtl433hip", SEAM_OUT)
    return OUT


def _build_variant(name, extra, verbose):
    out = os.path.join(OUT_DIR, name)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INC, f) for f in os.listdir(INC)]
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    hipcc = _hipcc()
    objs, procs = [], []
    tmp = os.path.join(OUT_DIR, "_" + name)
    os.makedirs(tmp, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(tmp, src.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc] + FLAGS + extra + ["-I", INC, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        o, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{o.decode(errors='replace')}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", out] + objs)
    shutil.rmtree(tmp)
    return out


def build_seam(lib_dir, lib_name, out):
    """librtl433seam.so: plain host C++ (no device code), linked against the library whose C ABI it wraps."""
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-I", INC, os.path.join(CSRC, "ref_seam.cpp"), "-o", out,
           "-L", lib_dir, "-l" + lib_name, "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib", "-lm"]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, timing="--timing" in sys.argv))
