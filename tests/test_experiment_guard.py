"""Timing-experiment builds of the library cannot reach a user's process by accident (round-4 verdict, weak point 8):
their switches only compile under -DP2P_EXPERIMENT, which marks p2p_version(); the binding refuses a marked library and the
P2P_LIB_PATH override unless P2P_ALLOW_EXPERIMENT=1."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "patch2pix_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _syntax_only(*defines):
    return subprocess.run([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", *defines, os.path.join(CSRC, "api.hip")],
                          capture_output=True, text=True, timeout=300)


def test_experiment_switch_without_marker_does_not_compile():
    r = _syntax_only("-DXF_PIN_W")
    assert r.returncode != 0 and "P2P_EXPERIMENT" in r.stderr
    assert _syntax_only("-DXF_PIN_W", "-DP2P_EXPERIMENT").returncode == 0
    assert _syntax_only().returncode == 0


def test_every_switch_in_the_sources_is_guarded():
    """Every XF_* / XH_* / XP_* / *_TIMING macro the kernels test appears in the guard of p2p_common.h."""
    import re
    guard = open(os.path.join(CSRC, "p2p_common.h")).read().split("#error")[0]
    used = set()
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h")):
            for line in open(os.path.join(CSRC, f)):
                if line.lstrip().startswith(("#if", "#ifdef", "#ifndef", "#elif")):
                    used.update(re.findall(r"\b(XF_[A-Z0-9_]+|XH_[A-Z0-9_]+|XP_[A-Z0-9_]+|[A-Z0-9_]+_TIMING|P2P_WINO_CHUNK)\b", line))
    used.discard("XF_WINO_STAGGER_US")      # the parameter of XF_WINO_STAGGER
    missing = sorted(m for m in used if f"defined({m})" not in guard)
    assert not missing, missing


def test_binding_refuses_library_override_without_consent():
    code = "import patch2pix_amd._lib"
    env = {k: v for k, v in os.environ.items() if k != "P2P_ALLOW_EXPERIMENT"}
    env["P2P_LIB_PATH"] = os.path.join(CSRC, "libp2p_hip.so")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "P2P_ALLOW_EXPERIMENT" in r.stderr
    env["P2P_ALLOW_EXPERIMENT"] = "1"
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
