"""Build libvallex.so (HIP, gfx950 only) in-tree next to this file.

    python vall-e_amd/csrc/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU; the resulting .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libvallex.so")
SRCS = ["engine.hip"]
# every header next to this file is a dependency (a hand-kept list went stale once: an edited header did not trigger a rebuild)
DEPS = SRCS + sorted(f for f in os.listdir(HERE) if f.endswith(".hpp")) + ["../../include/vallex.h", "build.py"]


def stale() -> bool:
    if not os.path.isfile(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
           "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
           # kernel arguments (the first 16 SGPRs' worth of explicit ones) arrive in registers at wave launch instead of by an
           # s_load the kernel's first instructions wait for: 0.3 us per dependent kernel of the decode step
           "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "-o", OUT] + [os.path.join(HERE, s) for s in SRCS]
    if os.environ.get("VX_SAVE_TEMPS"):
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


def build_stamps() -> str:
    """Probe build with in-kernel phase stamps (common.hpp VX_STAMP) -> libvallex_stamps.so; never loaded by the package."""
    out = os.path.join(HERE, "libvallex_stamps.so")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-DVX_STAMPS", "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "-Wno-unused-function", "-Wno-unused-variable", "-o", out] + [os.path.join(HERE, s) for s in SRCS]
    subprocess.check_call(cmd, cwd=HERE)
    return out


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build_stamps())
    else:
        build(force="--force" in sys.argv)
        print(OUT)
