"""TEST INFRASTRUCTURE: compile the kernel sources of patch2pix_amd/csrc for the HOST against the HIP stand-in in
this directory (tests/hipemu/hip/hip_runtime.h) -> tests/hipemu/_build/libp2p_emu.so, the same C ABI operating on
host pointers.  Only tests/test_kernels_emulated.py loads it; the product never does."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "patch2pix_amd", "csrc")
OUT = os.path.join(HERE, "_build", "libp2p_emu.so")
SOURCES = ["api.hip", "backbone.hip", "coarse.hip", "consensus.hip", "filter.hip", "regress.hip", "regress_h2.hip", "regress_wino.hip"]


def build(force=False, verbose=False):
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"),
             os.path.join(ROOT, "include", "p2p_hip.h")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cxx = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    cmd = [cxx, "-std=c++17", "-O2", "-g0", "-fPIC", "-shared", "-pthread", "-ffp-contract=off", "-I" + HERE,
           "-Wno-unused-value", "-Wno-unknown-attributes", "-o", OUT]
    cmd += os.environ.get("HIPEMU_DEFINES", "").split()          # e.g. -DXF_L3_16 to run an experiment build on the CPU
    for s in SOURCES:
        cmd += ["-x", "c++", os.path.join(CSRC, s)]
    cmd += ["-x", "c++", os.path.join(HERE, "hipemu.cpp")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))


def build_example(force=False, verbose=False):
    """examples/cabi_coarse.c compiled as C++ against the stand-in and linked to the emulated library: the same
    program as on the GPU box, runnable here."""
    lib = build(force=force, verbose=verbose)
    src = os.path.join(ROOT, "examples", "cabi_coarse.c")
    exe = os.path.join(HERE, "_build", "cabi_coarse_emu")
    if not force and os.path.exists(exe) and os.path.getmtime(exe) >= max(os.path.getmtime(src), os.path.getmtime(lib)):
        return exe
    cxx = os.environ.get("HIPEMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    cmd = [cxx, "-x", "c++", "-std=c++17", "-O1", "-I" + HERE, "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
           "-L" + os.path.dirname(lib), "-lp2p_emu", "-Wl,-rpath,$ORIGIN", "-pthread"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return exe
