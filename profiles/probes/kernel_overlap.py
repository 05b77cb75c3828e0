"""From a rocprofv3 --kernel-trace CSV: per kernel name the calls, the mean duration and the share of its time during which at least one OTHER
kernel was running too (streams side by side).  usage: kernel_overlap.py <kernel_trace.csv> [last_n_kernels]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Queue_Id", "")) for r in rows), key=lambda k: k[0])
ks = [k for k in ks if not k[2].startswith("__amd_rocclr")]
if len(sys.argv) > 2:
    ks = ks[-int(sys.argv[2]):]
stat = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, (s, e, n, q) in enumerate(ks):
    ov = 0
    # union of the other kernels' intervals clipped to [s, e]
    iv = sorted((max(s, s2), min(e, e2)) for j, (s2, e2, n2, q2) in enumerate(ks) if j != i and s2 < e and e2 > s)
    cur = s
    for a, b in iv:
        if b > cur:
            ov += b - max(a, cur)
            cur = b
    st = stat[n]
    st[0] += 1; st[1] += e - s; st[2] += ov
span = ks[-1][1] - ks[0][0]
print("kernels %d, span %.1f us, queues %d" % (len(ks), span / 1e3, len(set(k[3] for k in ks))))
print("%-50s %6s %10s %12s" % ("kernel", "calls", "mean_us", "overlapped_%"))
for n, (c, t, o) in sorted(stat.items(), key=lambda kv: -kv[1][1]):
    print("%-50s %6d %10.1f %12.1f" % (n, c, t / c / 1e3, 100.0 * o / t if t else 0.0))
