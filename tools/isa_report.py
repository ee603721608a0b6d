#!/usr/bin/env python3
"""ISA facts of the table kernels for one seat count, from hipcc's own assembly output with the library's build flags:
flat / global / scratch memory instructions, SGPR spill traffic (v_writelane / v_readlane), registers, occupancy, and the
VALU opcode histogram with the half-rate share bench.py's `ceiling_mix` uses (profiles/r03_valu_rates.txt: everything but
the plain two-operand 32-bit ALU kinds issues at half rate).
usage: tools/isa_report.py [N ...] [--json out.json]      (default N = 6)"""
import json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pokerl_amd import build

# plain full-rate VALU kinds (two-operand 32-bit integer ALU, moves, 32-bit compares): measured 2.25-2.5 cycles with >= 4 waves
FULL_RATE = re.compile(r"^v_(add|sub|subrev|and|or|xor|xnor|not|mov)_(u32|i32|b32)(_e32|_e64|_sdwa|_dpp)?$|"
                       r"^v_(add|sub|subrev|addc|subb|subbrev)_co(_ci)?_u32(_e32|_e64|_sdwa|_dpp)?$|"
                       r"^v_cmpx?_\w+_(u32|i32)(_e32|_e64|_sdwa)?$|^v_accvgpr")

def report(n):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "t.s")
        r = subprocess.run([build.hipcc()] + build.table_flags(n) + ["-DPK_SEATS=%d" % n, "--cuda-device-only", "-S",
                            "-Rpass-analysis=kernel-resource-usage", os.path.join(build.CSRC, "pk_tables.hip"), "-o", asm],
                           capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        s = open(asm).read()
    usage, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); usage[cur] = {}; continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z /\[\]]*?): (\S+) \[-Rpass", line)
        if m and cur:
            usage[cur][m.group(1).strip()] = m.group(2)
    out = {}
    parts = re.split(r"\n(_Z\w+):\s*; @", s)
    for i in range(1, len(parts), 2):
        mangled = parts[i]
        name = re.sub(r"\(.*", "", subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()).replace("void ", "")
        body = parts[i + 1].split(".Lfunc_end")[0]
        ops = re.findall(r"\n\s+([a-z_0-9]+)[ \n]", body)
        valu = [o for o in ops if o.startswith("v_") and not o.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))]
        hist = {}
        for o in valu:
            hist[o] = hist.get(o, 0) + 1
        half = sum(c for o, c in hist.items() if not FULL_RATE.match(o))
        u = usage.get(mangled, {})
        out[name] = dict(flat=sum(o.startswith("flat_") for o in ops), global_=sum(o.startswith("global_") for o in ops),
                         scratch=sum(o.startswith("scratch_") for o in ops), buffer=sum(o.startswith("buffer_") for o in ops),
                         writelane=ops.count("v_writelane_b32"), readlane=ops.count("v_readlane_b32"),
                         valu_static=len(valu), salu_static=sum(o.startswith("s_") for o in ops), lds_static=sum(o.startswith("ds_") for o in ops),
                         half_rate_share_static=round(half / max(1, len(valu)), 4),
                         vgprs=int(u.get("VGPRs", -1)), agprs=int(u.get("AGPRs", -1)), sgprs=int(u.get("TotalSGPRs", -1)),
                         sgpr_spill=int(u.get("SGPRs Spill", -1)), vgpr_spill=int(u.get("VGPRs Spill", -1)),
                         scratch_bytes=int(u.get("ScratchSize [bytes/lane]", -1)), occupancy=int(u.get("Occupancy [waves/SIMD]", -1)),
                         lds_bytes=int(u.get("LDS Size [bytes/block]", -1)),
                         top_valu=sorted(hist.items(), key=lambda kv: -kv[1])[:12])
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    seats = [int(a) for a in args if a.isdigit()] or [6]
    res = {}
    for n in seats:
        res[str(n)] = rep = report(n)
        print("---- %d seats" % n)
        print("%-28s %5s %6s %7s %5s %5s %5s %6s %6s %4s %5s %5s %5s" % ("kernel", "flat", "global", "scratch", "wlane", "rlane", "VGPR", "AGPR", "spillS", "occ", "VALU", "SALU", "half"))
        for k, d in rep.items():
            print("%-28s %5d %6d %7d %5d %5d %5d %6d %6d %4d %5d %5d %5.2f" % (k, d["flat"], d["global_"], d["scratch"], d["writelane"], d["readlane"], d["vgprs"], d["agprs"],
                                                                                 d["sgpr_spill"], d["occupancy"], d["valu_static"], d["salu_static"], d["half_rate_share_static"]))
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
