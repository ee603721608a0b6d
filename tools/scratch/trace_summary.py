import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_env_step_async" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 100000]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-1500:]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
qs = sorted(set(r.get("Queue_Id", "?") for r in rows))
print("launches %d  mean kernel %.1f us  span per launch %.1f us  overlap factor %.2f  queues %s" % (len(d), sum(d) / len(d) / 1e3, span / len(d) / 1e3, sum(d) / span, qs))
