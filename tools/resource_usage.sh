#!/bin/bash
# Register / spill / LDS / occupancy figures and the memory-instruction kinds (flat / global / scratch) of every table kernel
# of libpokerl_hip.so for every seat count, as the compiler reports them with the library's own build flags, plus the VALU
# opcode histogram's half-rate share that bench.py's `ceiling_mix` uses (tools/isa_report.py does the work; the seat counts
# are compiled in parallel).
# usage: tools/resource_usage.sh rNN     -> profiles/rNN_resource_usage.txt, profiles/rNN_isa_report.json
cd "$(dirname "$0")/.."
tag=${1:-r05}
seq 2 16 | xargs -P 8 -I{} sh -c 'python3 tools/isa_report.py {} --json /tmp/isa_report_{}.json > /tmp/isa_report_{}.txt'
for n in $(seq 2 16); do cat /tmp/isa_report_$n.txt; done > profiles/${tag}_resource_usage.txt
python3 - "$tag" <<'PY'
import json, sys
out = {}
for n in range(2, 17):   # build.SEATS
    out.update(json.load(open("/tmp/isa_report_%d.json" % n)))
json.dump(out, open("profiles/%s_isa_report.json" % sys.argv[1], "w"), indent=1)
PY
echo "profiles/${tag}_resource_usage.txt profiles/${tag}_isa_report.json"
