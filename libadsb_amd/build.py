"""Build the native libraries in-tree (they travel to the GPU box with the snapshot).

    python -m libadsb_amd.build            # everything
    python -m libadsb_amd.build --force

hipcc cross-compiles gfx950 code objects without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")

LIB = os.path.join(HERE, "libadsb_amd.so")
SYNTH = os.path.join(HERE, "libadsb_synth.so")

HIP_SOURCES = ["scan1090.hip", "scan2400.hip", "uat978.hip", "capi.cpp", "resolver1090.cpp", "transport.cpp", "adsb1090_gpu_handler.cpp", "uat978_host.cpp", "uat978_gpu_handler.cpp"]
HIP_DEPS = HIP_SOURCES + ["scan1090.h", "uat978.h", "rs978.h", "resolver1090.hpp", "decode1090.h", "transport.hpp", "scan_common.hip.h", "gather1090.hip.h", "diag.hip.h", os.path.join(ROOT, "include", "adsb_amd.h"),
                           os.path.join(ROOT, "include", "libadsb_iface.hpp")]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d if os.path.isabs(d) else os.path.join(CSRC, d)) > t for d in deps)


# -amdgpu-atomic-optimizer-strategy=None for the two scan kernels' translation units only: they issue their one-lane atomics themselves and
# read a ticket's value half a chunk after issuing it, and the compiler's wave-aggregation of atomics wraps them in a sequence that waits
# for the value on the spot.  The UAT kernels' many-lane atomics (compaction) want the aggregation: with the flag on the whole library the
# UAT step went from 0.52 to 0.67 ms.
NO_ATOMIC_AGGREGATION = {"scan1090.hip", "scan2400.hip"}


def build_hip(force=False, verbose=False):
    if not force and not _stale(LIB, HIP_DEPS):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # host side: x86-64-v3 (AVX2/FMA class: inline floor/round) with contraction off, so that the double arithmetic of the CPR and
    # field decoding rounds exactly like the expressions as written (the oracle is built the same way)
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wextra", "-march=x86-64-v3", "-ffp-contract=off",
              "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in HIP_SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        cmd = common + (["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"] if src in NO_ATOMIC_AGGREGATION else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


def build_synth(force=False):
    srcs = [os.path.join(CSRC, "synth1090.c"), os.path.join(CSRC, "synth978.c")]
    if not force and not _stale(SYNTH, srcs + [os.path.join(CSRC, "synth1090.h"), os.path.join(CSRC, "synth978.h")]):
        return SYNTH
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-pthread", "-o", SYNTH] + srcs + ["-lm"])
    return SYNTH


CXX_TEST = os.path.join(ROOT, "tests", "cpp", "test_1090_gpu")
CXX_TEST_978 = os.path.join(ROOT, "tests", "cpp", "test_978_gpu")


def build_cxx_test(force=False):
    """The C++ drop-in test drivers (they use the reference-shaped factories exported by libadsb_amd.so)."""
    for exe in (CXX_TEST, CXX_TEST_978):
        src = exe + ".cpp"
        if not force and not _stale(exe, [src, LIB, os.path.join(ROOT, "include", "libadsb_iface.hpp")]):
            continue
        subprocess.check_call(["g++", "-std=c++20", "-O2", "-I" + os.path.join(ROOT, "include"), "-o", exe, src,
                               "-L" + HERE, "-ladsb_amd", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"])
    return CXX_TEST


def build_all(force=False, verbose=False):
    libs = build_hip(force, verbose), build_synth(force)
    build_cxx_test(force)
    return libs


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
