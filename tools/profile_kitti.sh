#!/bin/bash
# Kernel + memory-copy trace of the KITTI stream bench (configs[4]): where a step's time goes on the GPU side.
# usage (GPU box): bash tools/profile_kitti.sh [precision] [batch]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/kitti_prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PREC=${1:-fp16}; B=${2:-4}
rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$OUT/trace" -o kitti -- python3 "$ROOT/bench.py" --workload kitti --precision "$PREC" --batch "$B" --steps 40 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
cat "$OUT/bench.json"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for pat, title in (("*kernel_stats.csv", "kernels"), ("*memory_copy_stats.csv", "copies")):
    for f in glob.glob(os.path.join(out, "trace", "**", pat), recursive=True):
        rows = list(csv.DictReader(open(f)))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        print("== %s: total %.1f ms over the run (%s)" % (title, tot / 1e6, os.path.basename(f)))
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
            print("  %9.1f us avg x %6s = %8.2f ms  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["TotalDurationNs"]) / 1e6, r["Name"][:90]))
PY
