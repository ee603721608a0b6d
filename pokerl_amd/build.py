"""Builds libpokerl_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc.  No torch, no JIT cache."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpokerl_hip.so")
HEADERS = [os.path.join(CSRC, "pk_device.hpp"), os.path.join(CSRC, "pk_kernels.hpp"), os.path.join(os.path.dirname(HERE), "include", "pokerl_hip.h")]
# -ffp-contract=off: numpy never fuses multiply-add, so neither may we (bit-exact f64 money).
# -amdgpu-sched-strategy=max-ilp: the table kernels run ONE wave per SIMD (65 536 tables = 1 024 waves), where issue is bound
#   by dependent-instruction latency (tools/microbench/valu_rates.hip: 8.5 cycles dependent vs 5 independent), so the
#   scheduler should chase ILP, not occupancy: +10 % on k_rollout<6> (19.6 -> 21.6 G env-steps/s).
# -fno-honor-nans: money is never NaN, so `x > m ? x : m` may become v_max_f64 (-4 % VALU); signed zeros stay honoured.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fno-fast-math", "-fno-honor-nans", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fPIC", "-shared",
         "-fgpu-rdc=0" if False else "-Wall", "-Wno-unused-function"]


SEATS = list(range(2, 17))   # one object per seat count (pk_tables.hip -DPK_SEATS=N), PK_MIN_PLAYERS .. PK_MAX_PLAYERS
OBJ = os.path.join(HERE, "_obj")
SOURCES = ["pk_api.hip", "pk_tables.hip"]
COMPILE_FLAGS = [f for f in FLAGS if f != "-shared"]
# -enable-post-misched=0 (no post-register-allocation scheduler pass) for the table kernels of the seat counts where it was MEASURED
#   to pay: six seats (k_rollout<6> +0.6 ... 0.9 %, the all-in kernel +2.5 %) and four (+1.5 %); at every other seat count the default
#   is faster (-0.2 ... -4.6 %, most at the wide tables) and the streaming evaluator in pk_api.hip loses 1.5 % with it.  A scheduling
#   lottery per kernel, so: picked per seat count like the occ3 kernels (profiles/r04_flag_variants.txt).
POST_MISCHED_OFF_SEATS = (4, 6)


def table_flags(seats):
    return COMPILE_FLAGS + (["-mllvm", "-enable-post-misched=0"] if seats in POST_MISCHED_OFF_SEATS else [])

def source_hash():
    """16 hex digits over what the kernels are compiled from: csrc/* and the ABI header with comments and white space removed (a reworded
    comment must not make every committed profile look stale), the compiler flags and the per-seat-count flag choices.  Embedded in the library
    (pk_build_info) and stamped into every profiles/*_summary.json, so that bench.py can tell whether a committed counter summary describes
    the library it is running (`profile_stale`)."""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [HEADERS[-1]]:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        if not path.endswith((".hip", ".hpp", ".h")):
            continue
        txt = open(path).read()
        txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
        txt = re.sub(r"//[^\n]*", " ", txt)
        h.update(os.path.basename(path).encode() + b"\0" + " ".join(txt.split()).encode() + b"\0")
    h.update(repr((FLAGS, POST_MISCHED_OFF_SEATS, SEATS)).encode())
    return h.hexdigest()[:16]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpokerl_hip.so cannot be built")
    return exe


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def _compile(job):
    src, obj, defines, verbose = job
    seats = [int(d.split("=")[1]) for d in defines if d.startswith("-DPK_SEATS=")]
    cmd = [hipcc()] + (table_flags(seats[0]) if seats else COMPILE_FLAGS) + defines + ["-c", os.path.join(CSRC, src), "-o", obj]
    if not seats:      # pk_api.hip carries the hash of ALL kernel sources (pk_build_info)
        cmd.insert(-4, '-DPK_SOURCE_HASH="%s"' % source_hash())
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build_variant(out, defines=(), seats=None, tag="", verbose=False, jobs=None, table_defines=()):
    """Compiles pk_api.hip and one pk_tables.hip object per seat count IN PARALLEL (the table kernels of one seat count take
    10-40 s of hipcc each; in one translation unit the library took three minutes), then links them.  seats: None = all of
    SEATS; one seat count = a development library that holds that seat count only (-DPK_ONLY_SEATS), built in seconds."""
    from concurrent.futures import ThreadPoolExecutor
    defines = list(defines)
    only = None
    if seats is not None:
        seats = list(seats)
        if len(seats) != 1:
            raise ValueError("seats: None (all) or exactly one seat count")
        only = seats[0]
        defines = defines + ["-DPK_ONLY_SEATS=%d" % only]
    else:
        seats = SEATS
    os.makedirs(OBJ, exist_ok=True)
    work = [("pk_api.hip", os.path.join(OBJ, "pk_api%s.o" % tag), defines, verbose)]
    # widest tables first: they take longest to compile
    work += [("pk_tables.hip", os.path.join(OBJ, "pk_tables_%d%s.o" % (n, tag)), defines + list(table_defines) + ["-DPK_SEATS=%d" % n], verbose)
             for n in sorted(seats, reverse=True)]
    jobs = jobs or int(os.environ.get("PK_BUILD_JOBS", "0")) or min(8, os.cpu_count() or 1)
    with ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(_compile, work))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build_prof_lib(verbose=False, seats=None):
    """Diagnostic build with s_memtime stamps around the step machine's blocks (tools/block_profile.py); not shipped."""
    return build_variant(os.path.join(HERE, "libpokerl_hip_prof.so"), ["-DPK_PROFILE"], seats=seats, tag="_prof", verbose=verbose)


def build_counts_lib(verbose=False, seats=None):
    """Diagnostic build that COUNTS wave-level events with global atomics (lanes served per end_block, side-pot passes and
    their active lanes); its timings are meaningless.  tools/block_profile.py with PK_COUNTS=1; not shipped."""
    return build_variant(os.path.join(HERE, "libpokerl_hip_counts.so"), ["-DPK_PROFILE", "-DPK_PROFILE_COUNTS"], seats=seats, tag="_counts", verbose=verbose)


def build_dev_lib(seats=6, defines=(), name=None, verbose=False):
    """A development library with ONE seat count (A/B of kernel variants: POKERL_HIP_LIB=<path> selects it); not shipped."""
    out = os.path.join(HERE, name or "libpokerl_hip_dev%d.so" % seats)
    return build_variant(out, list(defines), seats=[seats], tag="_dev%d%s" % (seats, "".join(d.replace("-D", "_").replace("=", "") for d in defines)), verbose=verbose)


def build_lib(force=False, verbose=False):
    if not force and not stale():
        return LIB
    return build_variant(LIB, verbose=verbose)


if __name__ == "__main__":
    import sys
    if "--dev" in sys.argv:
        n = int(sys.argv[sys.argv.index("--dev") + 1])
        print(build_dev_lib(n, [a for a in sys.argv if a.startswith("-D")], verbose=True))
        sys.exit(0)
    build_lib(force="--force" in sys.argv, verbose=True)
    if "--prof" in sys.argv:
        build_prof_lib(verbose=True)
    if "--counts" in sys.argv:
        build_counts_lib(verbose=True)
