#!/usr/bin/env python3
"""profiles/pmc_fir_mac.json FROM a committed rocprofv3 summary, never by hand ("make the line's evidence track the build").

    python profiles/make_pmc_json.py profiles/r05_bench_512ch_rocprof.txt            # rewrite profiles/pmc_fir_mac.json
    python profiles/make_pmc_json.py --check                                         # exit 1 if the json differs from the summary it names

The summary is what profiles/run_rocprof.sh writes (summarize_rocprof.py): the kernel table of `rocprofv3 --kernel-trace --stats` and the
"corrected HBM traffic per launch" lines of the two separate --pmc passes (read = 2 x FETCH_SIZE: the guide's gfx950 correction; write = WRITE_SIZE).
bench.py puts `traffic_bytes_per_launch` of the roofline kernel and of the segment kernel into its JSON line; tests/test_profiles_track_build.py
fails when this file does not follow its source or when the source is older than the kernels' sources in git history.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
JSON_PATH = os.path.join(ROOT, "profiles", "pmc_fir_mac.json")
# kernel names as the summary prints them -> key of the json
KERNELS = {
    "fir_inv_kernel<13, 1, false>": None,                 # the roofline kernel: top level
    "fir_inv_kernel<13, 1, true>": "chained_variant",
    "segf_kernel<0>": "segment_kernel",                   # one frame per launch (round 4's summaries print segf_kernel<false>)
    "fir_fwd13wh_kernel<0>": "forward_kernel",            # (round 4: fir_fwd13w_kernel<0>)
}
OLD_NAMES = {"segf_kernel<false>": "segf_kernel<0>", "fir_fwd13w_kernel<0>": "fir_fwd13wh_kernel<0>"}
DESCRIPTION = {
    None: "fir_inv_kernel<13, 1, false> (spectrum multiply-accumulate fused into the inverse FFT; the plain variant = amp 2 of the benchmark chain)",
    "chained_variant": "fir_inv_kernel<13, 1, true> (amp 1: + the forward transform of amp 2)",
    "segment_kernel": "segf_kernel<0> (seg.hip compiled with SEG_FAST: 512 threads, one LDS frame buffer, two workgroups per CU; average over the two segment launches of a step)",
    "forward_kernel": "fir_fwd13wh_kernel<0> (forward transform of amp 1's frame, one LDS buffer)",
}


def parse_summary(path):
    """{kernel: {"read_bytes", "write_bytes", "traffic_bytes_per_launch", "avg_us", "calls"}} from a summarize_rocprof.py text."""
    out = {}
    traffic = re.compile(r"^(\S.*?)\s+read\s+([0-9.]+) MB\s+write\s+([0-9.]+) MB\s+total\s+([0-9.]+) MB\s*$")
    table = re.compile(r"^(\S.*?)\s+(\d+)\s+(\d+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s*$")
    with open(path) as f:
        for line in f:
            m = traffic.match(line)
            if m:
                k = OLD_NAMES.get(m.group(1).strip(), m.group(1).strip())
                out.setdefault(k, {}).update(read_bytes=float(m.group(2)) * 1e6, write_bytes=float(m.group(3)) * 1e6,
                                             traffic_bytes_per_launch=float(m.group(4)) * 1e6)
                continue
            m = table.match(line)
            k = OLD_NAMES.get(m.group(1).strip(), m.group(1).strip()) if m else None
            if m and k in KERNELS and "avg_us" not in out.get(k, {}):
                out.setdefault(k, {}).update(calls=int(m.group(3)), avg_us=float(m.group(5)))
    return out


def build(summary_rel):
    s = parse_summary(os.path.join(ROOT, summary_rel))
    missing = [k for k in KERNELS if "traffic_bytes_per_launch" not in s.get(k, {})]
    if "fir_inv_kernel<13, 1, false>" in missing:
        raise SystemExit("%s has no counter lines for the roofline kernel" % summary_rel)
    doc = {
        "workload": "512ch x 8192 frames x 65536 taps (K=8), per-channel IRs",
        "workload_key": "512x8192x65536",
        "fused": True,
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/run_rocprof.sh); read = 2 x FETCH_SIZE KiB "
                  "(gfx950 correction, MI355X_MICROARCH.md HBM section), write = WRITE_SIZE KiB; written by profiles/make_pmc_json.py",
        "source": summary_rel,
    }
    for kernel, key in KERNELS.items():
        if kernel in missing:
            continue
        e = {"kernel": DESCRIPTION[key], "traffic_bytes_per_launch": s[kernel]["traffic_bytes_per_launch"],
             "read_bytes": s[kernel]["read_bytes"], "write_bytes": s[kernel]["write_bytes"]}
        if "avg_us" in s[kernel]:
            e["avg_launch_us_kernel_trace"] = s[kernel]["avg_us"]
        if key is None:
            doc.update(e)
        else:
            doc[key] = e
    return doc


def main():
    if len(sys.argv) == 2 and sys.argv[1] == "--check":
        with open(JSON_PATH) as f:
            have = json.load(f)
        want = build(have["source"])
        if have != want:
            sys.stderr.write("profiles/pmc_fir_mac.json does not follow %s: run python profiles/make_pmc_json.py %s\n" % (have["source"], have["source"]))
            return 1
        print("profiles/pmc_fir_mac.json follows %s" % have["source"])
        return 0
    if len(sys.argv) != 2:
        sys.stderr.write(__doc__)
        return 2
    rel = os.path.relpath(os.path.abspath(sys.argv[1]), ROOT)
    doc = build(rel)
    with open(JSON_PATH, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print("wrote profiles/pmc_fir_mac.json from %s: roofline kernel %.1f MB, segment kernel %.1f MB per launch"
          % (rel, doc["traffic_bytes_per_launch"] / 1e6, doc.get("segment_kernel", {}).get("traffic_bytes_per_launch", 0) / 1e6))
    return 0


if __name__ == "__main__":
    sys.exit(main())
