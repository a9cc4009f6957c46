"""TEST INFRASTRUCTURE: builds tests/emu/_build/librtl433emu.so -- the kernels and host code of
rtl_433_amd/csrc compiled by g++ against the lock-step wave emulator (tests/emu/include/hip/hip_runtime.h).
Same sources as the product, no GPU, no hipcc.  Used only by tests/test_emu_*.py.

    python -m tests.emu.build_emu [--force]
"""
from __future__ import annotations

import os
import platform
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "rtl_433_amd", "csrc")
INC = os.path.join(ROOT, "include")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "librtl433emu.so")
SEAM_OUT = os.path.join(OUT_DIR, "librtl433seam.so")  # csrc/ref_seam.cpp over the emulator library
FLAGS = ["-O2", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing", "-Wall", "-Wno-unused-function",
         "-Wno-unknown-pragmas", "-Wno-unused-variable", "-Wno-sign-compare"]


def available():
    return platform.machine() == "x86_64"


def sources():
    from rtl_433_amd.build import SOURCES
    return [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "hip_emu_rt.cpp"), os.path.join(HERE, "selftest_kernels.hip")]


def _stale():
    if not os.path.exists(OUT) or not os.path.exists(SEAM_OUT):
        return True
    t = min(os.path.getmtime(OUT), os.path.getmtime(SEAM_OUT))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INC, f) for f in os.listdir(INC)]
    deps += [os.path.join(HERE, "hip_emu_rt.cpp"), os.path.join(HERE, "selftest_kernels.hip"), os.path.join(HERE, "include", "hip", "hip_runtime.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False):
    if not force and not _stale():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    # one builder at a time: the two ranks of a gloo test that both find the library stale would otherwise write the same
    # object files and link half of each other's
    import fcntl
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale():  # (the other one built it while this one waited)
            return OUT
        return _build_locked()


def _build_locked():
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        cmd = ["g++"] + FLAGS + ["-x", "c++", "-I", os.path.join(HERE, "include"), "-I", INC, "-I", CSRC, "-c", src, "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"g++ (emulator build) failed on {src}:\n{out.decode(errors='replace')}")
    subprocess.check_call(["g++", "-shared", "-o", OUT] + objs)
    for o in objs:
        os.remove(o)
    from rtl_433_amd.build import build_seam
    build_seam(OUT_DIR, "rtl433emu", SEAM_OUT)
    return OUT


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    print(build(force="--force" in sys.argv))
