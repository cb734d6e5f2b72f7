#!/bin/bash
# rocprofv3 kernel stats of the Isensee 3-D training step (reference defaults).  usage: bash tools/prof_isensee.sh
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_isensee_r03
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o stats -- python3 $ROOT/tools/bench_isensee.py --steps 8 > $OUT/run.log 2>&1
cd $ROOT
tail -1 $OUT/run.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step (10 steps): %.3f ms" % (tot / 10 / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")[:64]
    print("%-64s calls %5s  avg %8.1f us  per step %7.3f ms  %5.1f%%" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 10 / 1e6, float(r["Percentage"])))
PY
