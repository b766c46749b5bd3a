#!/usr/bin/env python3
"""traffic_json.py -- profiles/pmc_traffic.json from the counter passes of tools/pmc_traffic.sh (gpurun_out/traffic/raw.txt).
usage: python tools/traffic_json.py <raw.txt> <bench.json of the same build> [commit] [kernel_stats.csv of rocprofv3 --stats]
Method (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE come from separate --pmc passes; on gfx950 FETCH_SIZE
tallies a wide coalesced stream at half its bytes, calibrated here on seed_reads_kernel (split pipeline), whose fetches are the
read bases streamed with 16-B lane loads; the missing share of the stream is added to the fused kernel's figure, everything
else (index slots, minimizer lists read back, Match scratch) is left as the counter reports it."""
import json
import subprocess
import sys

raw, bench = sys.argv[1], sys.argv[2]
commit = sys.argv[3] if len(sys.argv) > 3 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
v = {}
for ln in open(raw):
    mode, ctr, kern, val, n = ln.split()
    v[(mode, ctr, kern.split("<")[0])] = float(val)
j = json.loads(open(bench).read().strip().splitlines()[-1])
streamed = j["config"]["bases_per_step_per_gpu"]
fetch = v[("fused", "FETCH_SIZE", "map_kernel")] * 1024
write = v[("fused", "WRITE_SIZE", "map_kernel")] * 1024
cal = v[("split", "FETCH_SIZE", "seed_reads_kernel")] * 1024 / streamed
total = fetch + streamed * (1.0 - cal) + write
rocprof = None
if len(sys.argv) > 4:
    import csv
    for row in csv.DictReader(open(sys.argv[4])):
        if row["Name"].startswith("void map_kernel<64, false, false>"):
            rocprof = {"calls": int(row["Calls"]), "average_ms": round(float(row["AverageNs"]) / 1e6, 4), "min_ms": round(float(row["MinNs"]) / 1e6, 4),
                       "max_ms": round(float(row["MaxNs"]) / 1e6, 4)}
out = {
    "reads": j["config"]["reads_per_step_per_gpu"],
    "genome_scale": 1.0,
    "hbm_bytes_per_launch": int(total),
    "commit": commit,
    "kernel": "map_kernel",
    "FETCH_SIZE_KB": v[("fused", "FETCH_SIZE", "map_kernel")],
    "WRITE_SIZE_KB": v[("fused", "WRITE_SIZE", "map_kernel")],
    "stream_calibration_ratio": round(cal, 4),
    "streamed_bytes": streamed,
    "split_pipeline_KB": {"seed_reads_FETCH": v[("split", "FETCH_SIZE", "seed_reads_kernel")], "seed_reads_WRITE": v[("split", "WRITE_SIZE", "seed_reads_kernel")],
                          "map_lists_FETCH": v[("split", "FETCH_SIZE", "map_lists_kernel")], "map_lists_WRITE": v[("split", "WRITE_SIZE", "map_lists_kernel")]},
    "algorithmic_bytes_per_launch": j["roofline"]["algorithmic_bytes_per_launch"],
    "map_kernel_rocprofv3": rocprof,
    "bench_event_avg_launch_ms": j["roofline"]["avg_launch_ms"],
    "algorithmic_GBps_at_rocprof_average": round(j["roofline"]["algorithmic_bytes_per_launch"] / (rocprof["average_ms"] * 1e-3) / 1e9, 1) if rocprof else None,
    "traffic_over_algorithmic": round(total / j["roofline"]["algorithmic_bytes_per_launch"], 3),
    "method": __doc__.split("Method ")[1].replace("\n", " "),
}
print(json.dumps(out, indent=1))
