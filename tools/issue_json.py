#!/usr/bin/env python3
"""issue_json.py -- profiles/pmc_issue.json from the counter passes of tools/pmc_kernels.sh (gpurun_out/<tag>/pmc).
usage: python tools/issue_json.py <pmc dir> <bench.json of the same build> <ceiling cycles per instruction> <ceiling source> [commit]
What bench.py's roofline.secondary needs ("Honest secondary bound: 64-bit integer VALU ... both must be reported"):
wave-instructions per map_kernel launch (SQ_INSTS_VALU + SALU + LDS + VMEM_RD + VMEM_WR + SMEM: each in its own --pmc pass group,
--kernel-trace only) and the shader clock of the run (GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = busy cycles of the
launch, over the launch's duration in the same pass's kernel trace)."""
import csv
import glob
import json
import re
import subprocess
import sys

pmc, bench, ceiling, source = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
commit = sys.argv[5] if len(sys.argv) > 5 else subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
KERN = "map_kernel<64, false, false>"
v = {}
for ln in open(pmc + "/summary.txt"):
    m = re.match(r"(\S.*?)\s{2,}(\S+)\s+per-launch\s+(\S+)", ln)
    if m and m.group(1).strip() == KERN:
        v[m.group(2)] = float(m.group(3))
# duration of map_kernel in the pass that counted GRBM_GUI_ACTIVE
dur = []
for f in glob.glob(pmc + "/pass*/**/*kernel_trace.csv", recursive=True):
    cc = glob.glob(f.rsplit("/", 1)[0] + "/*counter_collection.csv")
    if not cc or "GRBM_GUI_ACTIVE" not in open(cc[0]).read():
        continue
    for row in csv.DictReader(open(f)):
        if row.get("Kernel_Name", "").startswith("void " + KERN) or row.get("Kernel_Name", "").startswith(KERN):
            dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9)
j = json.loads(open(bench).read().strip().splitlines()[-1])
insts = sum(v.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"))
cycles = v["GRBM_GUI_ACTIVE"] / 8.0
n_simd = 256 * 4
mean_dur = sum(dur) / len(dur) if dur else None
out = {
    "reads": j["config"]["reads_per_step_per_gpu"],
    "genome_scale": 1.0,
    "k": int(re.search(r"k=(\d+)", j["metric"]).group(1)),
    "commit": commit,
    "kernel": "map_kernel",
    "wave_instructions_per_launch": int(insts),
    "valu": int(v.get("SQ_INSTS_VALU", 0)), "salu": int(v.get("SQ_INSTS_SALU", 0)), "lds": int(v.get("SQ_INSTS_LDS", 0)),
    "vmem": int(v.get("SQ_INSTS_VMEM_RD", 0) + v.get("SQ_INSTS_VMEM_WR", 0)), "smem": int(v.get("SQ_INSTS_SMEM", 0)),
    "busy_cycles_per_launch": int(cycles),
    "launch_s_in_counter_pass": mean_dur,
    "shader_clock_mhz": round(cycles / mean_dur / 1e6, 1) if mean_dur else 2400.0,
    "n_simd": n_simd,
    "waves_per_simd": 4,
    "cycles_per_instruction_in_counter_pass": round(cycles * n_simd / insts, 3),
    "ceiling_cycles_per_instruction": ceiling,
    "ceiling_source": source,
    "wait_share": {k: round(v[k] / v["SQ_WAVE_CYCLES"], 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in v and "SQ_WAVE_CYCLES" in v},
    "lds_bank_conflict_share": round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 3) if "SQ_LDS_IDX_ACTIVE" in v else None,
    "method": __doc__.split("What ")[1].replace("\n", " "),
}
print(json.dumps(out, indent=1))
