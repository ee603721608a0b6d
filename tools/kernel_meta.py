#!/usr/bin/env python3
"""Kernel descriptors of a BUILT library: private segment (scratch) bytes, registers, LDS of every kernel in the gfx950 code
objects embedded in libpokerl_hip.so (the `.hip_fatbin` offload bundles, one per translation unit), read from the AMDGPU metadata
notes with llvm-readelf.  What the kernels were compiled to, not what a re-compile would give.
usage: tools/kernel_meta.py [lib.so]          -> one line per kernel
CPU test: tests/test_host_api.py::test_no_table_kernel_uses_scratch (every table kernel: 0 bytes, except an explicit allow-list)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _readelf():
    for p in ("/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/llvm/bin/llvm-readelf"):
        if os.path.exists(p):
            return p
    raise RuntimeError("llvm-readelf not found under /opt/rocm")


def code_objects(lib_path):
    """The gfx950 code objects (bytes) of every offload bundle in the library."""
    data = open(lib_path, "rb").read()
    out = []
    pos = data.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(MAGIC))
        q = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "gfx950" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos = data.find(MAGIC, pos + 1)
    return out


def kernels(lib_path):
    """{demangled kernel name: dict(private_segment, vgprs, sgprs, agprs, lds, sgpr_spill, vgpr_spill)} over all code objects."""
    res = {}
    fields = {"private_segment": ".private_segment_fixed_size", "vgprs": ".vgpr_count", "sgprs": ".sgpr_count", "agprs": ".agpr_count",
              "lds": ".group_segment_fixed_size", "sgpr_spill": ".sgpr_spill_count", "vgpr_spill": ".vgpr_spill_count"}
    with tempfile.TemporaryDirectory() as td:
        for i, co in enumerate(code_objects(lib_path)):
            f = os.path.join(td, "co%d.o" % i)
            open(f, "wb").write(co)
            txt = subprocess.run([_readelf(), "--notes", f], capture_output=True, text=True, check=True).stdout
            # one block per kernel: "  - .agpr_count: ..." up to the next "  - " at that indent inside amdhsa.kernels
            m = re.search(r"amdhsa\.kernels:\n(.*?)\n\s*amdhsa\.target:", txt, re.S)
            if not m:
                continue
            for block in re.split(r"\n\s{2}- (?=\.)", "\n" + m.group(1)):
                name = re.search(r"\.name:\s+(\S+)", block)
                if not name:
                    continue
                d = {}
                for k, key in fields.items():
                    v = re.search(re.escape(key) + r":\s+(\d+)", block)
                    d[k] = int(v.group(1)) if v else -1
                res[name.group(1)] = d
    names = list(res)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return {re.sub(r"^void ", "", re.sub(r"\(.*", "", dem[i])) if i < len(dem) and dem[i] else n: res[n] for i, n in enumerate(names)}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pokerl_amd", "libpokerl_hip.so")
    ks = kernels(lib)
    print("%-44s %8s %5s %5s %5s %6s %7s %7s" % ("kernel", "scratchB", "VGPR", "AGPR", "SGPR", "LDS", "spillS", "spillV"))
    for k in sorted(ks, key=lambda s: (re.sub(r"<.*", "", s), int(re.search(r"<(\d+)", s).group(1)) if re.search(r"<(\d+)", s) else 0, s)):
        d = ks[k]
        print("%-44s %8d %5d %5d %5d %6d %7d %7d" % (k, d["private_segment"], d["vgprs"], d["agprs"], d["sgprs"], d["lds"], d["sgpr_spill"], d["vgpr_spill"]))
