#!/bin/bash
# Runs ON THE GPU BOX: tools/eval7_reconcile.py plain and under rocprofv3 --kernel-trace; then compares the profiler's timestamps of the same
# dispatches with the HIP-event figures.  usage: tools/eval7_reconcile.sh [out.txt]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
out=${1:-$ROOT/gpurun_out/r06/eval7_reconcile.txt}
mkdir -p "$(dirname "$out")"
D=$ROOT/gpurun_out/prof_r06_eval7_reconcile
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/eval7_reconcile.py 28 20 > $D/plain.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $ROOT/tools/eval7_reconcile.py 28 20 > $D/traced.json 2>$D/traced.err
python3 - $D > "$out" <<'PY'
import csv, glob, json, sys
d = sys.argv[1]
plain = json.loads(open(d + "/plain.json").read().strip().splitlines()[-1])
traced = json.loads([l for l in open(d + "/traced.json").read().splitlines() if l.startswith("{")][-1])
rows = [r for r in csv.DictReader(open(glob.glob(d + "/trace/*/*_kernel_trace.csv")[0])) if "k_eval7_tab_stream" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
print("HIP events, no profiler attached : back to back %s ms per pass; isolated launches %s" % (["%.4f" % x for x in plain["back_to_back_ms_per_pass"]], {k: round(v, 4) for k, v in plain["isolated_ms"].items()}))
print("HIP events, under rocprofv3      : back to back %s ms per pass; isolated launches %s" % (["%.4f" % x for x in traced["back_to_back_ms_per_pass"]], {k: round(v, 4) for k, v in traced["isolated_ms"].items()}))
i = 0
for gi, n in enumerate(traced["launch_groups"]):
    g = list(range(i, i + n)); i += n
    if not g or g[-1] >= len(rows):
        break
    durs = [(en[k] - st[k]) / 1e6 for k in g]
    span = (en[g[-1]] - st[g[0]]) / 1e6
    ov = [(en[g[k]] - st[g[k + 1]]) / 1e6 for k in range(len(g) - 1)]
    print("rocprofv3 group %d (%2d dispatch%s): kernel duration avg %.4f min %.4f max %.4f ms | first start -> last end %.4f ms = %.4f per pass | overlap of consecutive dispatches avg %s ms"
          % (gi, n, "es" if n > 1 else "  ", sum(durs) / n, min(durs), max(durs), span, span / n, ("%.4f" % (sum(ov) / len(ov))) if ov else "-"))
print("dispatches traced: %d" % len(rows))
PY
cat "$out"
