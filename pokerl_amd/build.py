"""Builds libpokerl_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.  No torch, no JIT cache."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpokerl_hip.so")
SOURCES = ["pk_api.hip"]
HEADERS = [os.path.join(CSRC, "pk_device.hpp"), os.path.join(CSRC, "pk_kernels.hpp"), os.path.join(os.path.dirname(HERE), "include", "pokerl_hip.h")]
# -ffp-contract=off: numpy never fuses multiply-add, so neither may we (bit-exact f64 money).
# -amdgpu-sched-strategy=max-ilp: the table kernels run ONE wave per SIMD (65 536 tables = 1 024 waves), where issue is bound
#   by dependent-instruction latency (tools/microbench/valu_rates.hip: 8.5 cycles dependent vs 5 independent), so the
#   scheduler should chase ILP, not occupancy: +10 % on k_rollout<6> (19.6 -> 21.6 G env-steps/s).
# -fno-honor-nans: money is never NaN, so `x > m ? x : m` may become v_max_f64 (-4 % VALU); signed zeros stay honoured.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fno-fast-math", "-fno-honor-nans", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC", "-shared",
         "-fgpu-rdc=0" if False else "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpokerl_hip.so cannot be built")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_prof_lib(verbose=False):
    """Diagnostic build with s_memtime stamps around the step machine's blocks (tools/block_profile.py); not shipped."""
    out = os.path.join(HERE, "libpokerl_hip_prof.so")
    cmd = [hipcc()] + FLAGS + ["-DPK_PROFILE"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_counts_lib(verbose=False):
    """Diagnostic build that COUNTS wave-level events with global atomics (lanes served per end_block, side-pot passes and
    their active lanes); its timings are meaningless.  tools/block_profile.py with PK_COUNTS=1; not shipped."""
    out = os.path.join(HERE, "libpokerl_hip_counts.so")
    cmd = [hipcc()] + FLAGS + ["-DPK_PROFILE", "-DPK_PROFILE_COUNTS"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def build_lib(force=False, verbose=False):
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    build_lib(force="--force" in sys.argv, verbose=True)
    if "--prof" in sys.argv:
        build_prof_lib(verbose=True)
    if "--counts" in sys.argv:
        build_counts_lib(verbose=True)
