"""Build libvallex.so (HIP, gfx950 only) in-tree next to this file.

    python vall-e_amd/csrc/build.py [--force]      product library (the C ABI of include/vallex.h, nothing else)
    python vall-e_amd/csrc/build.py --probes       libvallex_probes.so: product + the vx_debug_* measurement probes (probes.h)
    python vall-e_amd/csrc/build.py --stamps       libvallex_stamps.so: probes + in-kernel time stamps (-DVX_STAMPS)

hipcc cross-compiles for gfx950 without a GPU; the resulting .so travels to the GPU box with the
repo snapshot (it is git-ignored, not gpurun-ignored).  The package (engine.py) loads libvallex.so only."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "libvallex.so")
SRCS = ["engine.hip"]
# every header next to this file is a dependency (a hand-kept list went stale once: an edited header did not trigger a rebuild)
DEPS = SRCS + sorted(f for f in os.listdir(HERE) if f.endswith((".hpp", ".h"))) + ["../../include/vallex.h", "build.py"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
         # kernel arguments (the first 16 SGPRs' worth of explicit ones) arrive in registers at wave launch instead of by an
         # s_load the kernel's first instructions wait for: 0.3 us per dependent kernel of the decode step
         "-mllvm", "-amdgpu-kernarg-preload-count=16",
         # MFMA accumulators in VGPRs wherever the kernel does not need the AGPR file for capacity: the flash-attention loop spent
         # 128 of its ~280 VALU slots per key tile on v_accvgpr_read / _write copies (batched NAR stages 110.9 -> 107.7 ms)
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def stale(out: str = OUT) -> bool:
    if not os.path.isfile(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in DEPS)


def _compile(out: str, defines, verbose: bool) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [f"-D{d}" for d in defines] + ["-o", out] + [os.path.join(HERE, s) for s in SRCS]
    if os.environ.get("VX_SAVE_TEMPS"):
        cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    if verbose:  # to stderr: bench.py's stdout carries exactly one JSON line, and a stale library is rebuilt from there too
        print(" ".join(cmd), file=sys.stderr, flush=True)
    subprocess.check_call(cmd, cwd=HERE)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return OUT
    return _compile(OUT, [], verbose)


def build_probes(force: bool = False, verbose: bool = True) -> str:
    """Product code + the vx_debug_* probes (probes.h): never loaded by the package."""
    out = os.path.join(HERE, "libvallex_probes.so")
    return out if not force and not stale(out) else _compile(out, ["VX_PROBES"], verbose)


def build_stamps(force: bool = False, verbose: bool = True) -> str:
    """Probe build with in-kernel time stamps (common.hpp VX_STAMP / VX_KSTAMP): never loaded by the package."""
    out = os.path.join(HERE, "libvallex_stamps.so")
    return out if not force and not stale(out) else _compile(out, ["VX_PROBES", "VX_STAMPS"], verbose)


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--stamps" in sys.argv:
        print(build_stamps(force))
    elif "--probes" in sys.argv:
        print(build_probes(force))
    else:
        print(build(force))
