#!/bin/bash
# rocprofv3 kernel stats of a reference variant's training step (one stream: exclusive durations).  usage: bash tools/prof_variant.sh batch_norm
V=${1:-batch_norm}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$V
mkdir -p $OUT
export TMPDIR=/tmp ONE_STREAM=1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 $ROOT/tools/step_variant.py $V 8 > $OUT/run.log 2>&1
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (11 steps): %.3f ms" % (tot / 11 / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")[:70]
    print("%-70s calls %5s  avg %8.1f us  per step %7.3f ms  %5.1f%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 11 / 1e6, float(r["Percentage"])))
PY
