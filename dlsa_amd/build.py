"""Builds dlsa_amd/lib/libsw_hip.so (HIP kernels + C ABI) for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the .so is kept
in-tree (git-ignored) so that it travels to the GPU box with the source snapshot.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsw_hip.so")
SOURCES = ["sw_exact.hip", "sw_lanes.hip", "sw_score.hip", "sw_topk.hip", "sw_lib.cpp", "sw_lanes.cpp", "sw_db.cpp", "sw_node.cpp", "sw_package.cpp", "sw_worker.cpp"]
HEADERS = [os.path.join(CSRC, "sw_internal.h"), os.path.join(CSRC, "sw_host.h"), os.path.join(CSRC, "sw_score_asm.inc"), os.path.join(CSRC, "gen_score_asm.py"), os.path.join(HERE, "..", "include", "sw.h"),
           os.path.join(HERE, "..", "include", "sw_hip.h"), os.path.join(HERE, "..", "include", "sw_worker.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    m = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > m for d in deps)


# per-source extra flags.  sw_score.hip: keep wave-uniform branches as real branches (see the comment at its
# row dispatch); the flag is NOT applied to the other sources (it miscompiled a uniform loop nested in a
# divergent region of sw_exact.hip).
EXTRA = {"sw_score.hip": ["-mllvm", "-structurizecfg-skip-uniform-regions=true"]}


def check_generated():
    """sw_score_asm.inc is generated and committed: a generator that was edited without regenerating must not
    ship a stale row loop."""
    subprocess.check_call([sys.executable, os.path.join(CSRC, "gen_score_asm.py"), "--check"])


def build(force=False, verbose=False):
    check_generated()
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-x", "hip",
               os.path.join(CSRC, src), "-o", obj] + EXTRA.get(src, [])
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
