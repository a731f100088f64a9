"""Build libp2p_hip.so (gfx950) in-tree with hipcc.  `python -m patch2pix_amd.build`."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["api.hip", "backbone.hip", "coarse.hip", "consensus.hip", "filter.hip", "regress.hip", "regress_h2.hip", "regress_wino.hip"]
LIB = os.path.join(CSRC, "libp2p_hip.so")


STAMP = LIB + ".srchash"       # hash of the sources the library was built from (travels with the .so, git-ignored)


def _deps():
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "p2p_hip.h"))
    return deps


def source_hash():
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        h.update(open(d, "rb").read())
    return h.hexdigest()


def _stale():
    """By content, not by time stamp: a copied or checked-out tree does not keep mtimes, and a library older than the
    sources next to it must never be picked up silently."""
    if not os.path.exists(LIB):
        return True
    if os.path.exists(STAMP):
        return open(STAMP).read().strip() != source_hash()
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def _resource_report(text):
    """Parse -Rpass-analysis=kernel-resource-usage remarks into {kernel: {field: value}}."""
    report, cur = {}, None
    for line in text.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass")[0].strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            report[cur] = {}
        elif cur and ":" in body:
            k, v = body.rsplit(":", 1)
            report[cur][k.strip()] = v.strip()
    return report


def build(force=False, verbose=True):
    """Compile every HIP source for gfx950 into one shared library (no torch / pybind dependency)."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    headers = [d for d in _deps() if d.endswith(".h")]
    # the compiler is part of an object's identity: after a ROCm / HIPCC switch stale objects must not be relinked
    try:
        compiler_id = hipcc + "\n" + subprocess.run([hipcc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                                    text=True).stdout
    except OSError as e:
        compiler_id = hipcc + "\n" + repr(e)

    def compile_one(src):
        """One translation unit; skipped when the object on disk was built from the same source + headers + flags."""
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
               "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        h = hashlib.sha256(" ".join(cmd[1:-3]).encode())
        h.update(compiler_id.encode())
        for d in [os.path.join(CSRC, src)] + headers:
            h.update(os.path.basename(d).encode())
            h.update(open(d, "rb").read())
        stamp, key = obj + ".srchash", h.hexdigest()
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == key:
            return obj, ""
        if os.path.exists(stamp):
            os.remove(stamp)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        lines = [" ".join(cmd)] if verbose else []
        report = _resource_report(res.stdout)
        other = [l for l in res.stdout.splitlines() if "-Rpass-analysis" not in l and "remark:" not in l and l.strip()
                 and "|" not in l[:12] and not l.lstrip().startswith("^") and "__global__" not in l]
        if other and (verbose or res.returncode):
            lines += other
        if res.returncode:
            print("\n".join(lines), flush=True)
            raise subprocess.CalledProcessError(res.returncode, cmd)
        for name, usage in report.items():
            if verbose:
                lines.append(f"    {name}: {usage.get('VGPRs', '?')} VGPR + {usage.get('AGPRs', '?')} AGPR, "
                             f"scratch {usage.get('ScratchSize [bytes/lane]', '?')} B/lane")
            # Register spills are a hard error: kernels that touch scratch memory produced wrong results /
            # memory faults on the MI355X boxes (hipcc 7.2 code objects under torch's bundled HIP 7.0 runtime).
            if int(usage.get("ScratchSize [bytes/lane]", "0")) != 0:
                print("\n".join(lines), flush=True)
                raise RuntimeError(f"{src}: kernel {name} spills to scratch ({usage}); restructure it until it does not")
        with open(stamp, "w") as f:
            f.write(key)
        return obj, "\n".join(lines)

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(compile_one, SOURCES))
    objs = [o for o, _ in results]
    for _, text in results:
        if text:
            print(text, flush=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(source_hash())
    return LIB


ROOT = os.path.dirname(os.path.dirname(CSRC))
EXAMPLE_SRC = os.path.join(ROOT, "examples", "cabi_coarse.c")
EXAMPLE_BIN = os.path.join(ROOT, "examples", "_build", "cabi_coarse")


def build_examples(force=False, verbose=True, trust_existing=False):
    """examples/cabi_coarse.c: the C ABI used from plain C (gcc, C11) -- proves the header is C and the library
    needs nothing from Python.  Linked against the in-tree library with a relative rpath.
    trust_existing: use a binary that is already there without looking at time stamps (a copied tree -- the GPU box --
    does not necessarily keep them, and re-linking the shared library under a process that has it loaded is not an
    option)."""
    if trust_existing and not force and os.path.exists(EXAMPLE_BIN) and os.path.exists(LIB):
        return EXAMPLE_BIN
    if not os.path.exists(LIB) or not trust_existing:
        build(force=False, verbose=verbose)
    if (not force and os.path.exists(EXAMPLE_BIN) and os.path.getmtime(EXAMPLE_BIN) >= max(
            os.path.getmtime(EXAMPLE_SRC), os.path.getmtime(LIB))):
        return EXAMPLE_BIN
    os.makedirs(os.path.dirname(EXAMPLE_BIN), exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CC", "gcc"), "-std=c11", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include",
           "-I" + os.path.join(ROOT, "include"), EXAMPLE_SRC, "-o", EXAMPLE_BIN, "-L" + CSRC, "-lp2p_hip",
           f"-L{rocm}/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN/../../patch2pix_amd/csrc", f"-Wl,-rpath,{rocm}/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return EXAMPLE_BIN


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_examples(force="--force" in sys.argv)
