#!/bin/bash
# Code bytes per kernel and, inside map_kernel<64, false, false>, per stage (between the stage stamps of mq_clk): tools/code_sizes.sh > profiles/rNN_code_sizes.txt
# hipcc cross-compiles: no GPU needed.  The per-stage split comes from a diagnostic build (-DMQ_CODE_MARKS: every stamp is a symbol); the
# compiler lays a function's blocks out in roughly source order, cold blocks moved behind -- a map of where the bytes are, not an exact ledger.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B="/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-device-only -c -O3 -std=c++17 -Wno-unused-value -Wno-align-mismatch -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -pragma-unroll-threshold=65536"
T=$(mktemp -d)
$B -o $T/prod.o $ROOT/mapquik_amd/csrc/mq_capi.hip 2>/dev/null
$B -DMQ_CODE_MARKS -o $T/marks.o $ROOT/mapquik_amd/csrc/mq_capi.hip 2>/dev/null
for f in prod marks; do /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/$f.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$f.hsaco; done
echo "# code bytes per kernel (product build)"
/opt/rocm/lib/llvm/bin/llvm-readelf -sW $T/prod.hsaco | awk '$4=="FUNC"{print $3, $8}' | sort -n -r | uniq | c++filt | awk '{n=$1; $1=""; printf "%8d %s\n", n, $0}' | head -40
echo
echo "# map_kernel<64, false, false>: bytes between consecutive stage stamps, in address order (-DMQ_CODE_MARKS build; stamp numbers = mq_clk stage + 1:"
echo "#   0 wave start  1 stage A done  2 stage B done  3 stage R done  4 tile carry  5 list stores acknowledged  6 list in LDS  7 tuple hashes + probe issue  8 probes resolved + runs"
echo "#   9 runs finished  10 chain + result  11 general / declined  12 next work item)"
python3 - $T/marks.hsaco <<'PY'
import subprocess, sys, re
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-sW", sys.argv[1]], capture_output=True, text=True).stdout
syms = []
for ln in out.splitlines():
    p = ln.split()
    if len(p) >= 8 and re.match(r"^[0-9a-f]+$", p[1]):
        syms.append((int(p[1], 16), int(p[2]) if p[2].isdigit() else 0, p[3], p[7]))
k = [s for s in syms if s[3] == "_Z10map_kernelILi64ELb0ELb0EEv9SplitArgs" and s[2] == "FUNC"]
if not k:
    sys.exit("map_kernel<64, false, false> not found")
a0, size = k[0][0], k[0][1]
marks = sorted((a, n) for a, _, _, n in syms if n.startswith("mq_mark_") and a0 <= a < a0 + size)
print("kernel: %d bytes; %d stamps inside" % (size, len(marks)))
prev, prev_name = a0, "kernel entry"
tot = {}
for a, n in marks + [(a0 + size, "kernel end")]:
    stage = n.split("_")[2] if n.startswith("mq_mark_") else n
    print("  %7d bytes  %-14s -> stamp %s" % (a - prev, prev_name, stage))
    tot[stage] = tot.get(stage, 0) + (a - prev)
    prev, prev_name = a, "stamp " + stage
print("bytes in front of each stamp, summed over its copies:", dict(sorted(tot.items(), key=lambda kv: (len(kv[0]), kv[0]))))
PY
rm -rf $T
