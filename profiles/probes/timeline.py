"""One step's kernel timeline out of a rocprofv3 --kernel-trace CSV: the last `n` kernels (runtime fills / copies left out) with start offsets,
durations and queue ids.  usage: timeline.py <kernel_trace.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r.get("Queue_Id", "?")) for r in rows), key=lambda k: k[0])
ks = [k for k in ks if not k[2].startswith("__amd_rocclr")][-n:]
t0 = ks[0][0]
for s, e, name, q in ks:
    print("%9.1f us  +%7.1f us  queue %-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))
