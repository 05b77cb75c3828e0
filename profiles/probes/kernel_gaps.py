"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV: for every kernel the gap between the end of its predecessor (in
start order, all queues; runtime fill / copy kernels left out) and its own start.  usage: kernel_gaps.py <kernel_trace.csv> [skip_first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]) for r in rows), key=lambda k: k[0])
ks = [k for k in ks if not k[2].startswith("__amd_rocclr")][skip:]
gap_by = collections.defaultdict(list)
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(ks, ks[1:]):
    gap_by[(n0, n1)].append((s1 - e0) / 1e3)
for s, e, n in ks:
    busy += e - s
span = ks[-1][1] - ks[0][0]
print("kernels %d, span %.1f us, busy %.1f us (%.1f %%)" % (len(ks), span / 1e3, busy / 1e3, 100.0 * busy / span))
print("%-62s -> %-62s %6s %9s %9s" % ("after", "before", "n", "mean_us", "total_us"))
for (a, b), g in sorted(gap_by.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print("%-62s -> %-62s %6d %9.2f %9.1f" % (a, b, len(g), sum(g) / len(g), sum(g)))
