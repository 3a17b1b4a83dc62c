#!/bin/bash
# one per-launch timeline: tools/collect_timeline_one.sh <precision> <batch> <out.txt>   (rocprofv3 kernel trace of tools/profile_forward.py)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
prec="$1"; batch="$2"; out="$3"
D="$ROOT/gpurun_out/tl/prof_${prec}_b${batch}"
mkdir -p "$D"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$D" -o p --output-format csv -- python3 "$ROOT/tools/profile_forward.py" --precision "$prec" --batch "$batch" > "$D.log" 2>&1
f=$(find "$D" -name "*kernel_trace.csv" | head -1)
python3 "$ROOT/tools/timeline.py" "$f" --periodic --full > "$out"
cp "$(find "$D" -name "*kernel_stats.csv" | head -1)" "${out%.txt}_kernel_stats.csv" 2>/dev/null
rm -rf "$D"
