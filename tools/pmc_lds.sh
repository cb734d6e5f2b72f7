#!/bin/bash
# LDS bank conflicts per kernel from the counters (SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles; MI355X guide,
# LDS section): one training step under rocprofv3 --pmc (counters only, program directly behind `--`), aggregated per kernel name.
#   gpurun -- 'bash tools/pmc_lds.sh'   -> gpurun_out/pmc_lds.json
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc_lds
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT" -o lds -- python3 "$ROOT/bench.py" --steps 1 --warmup 1 --prewarm-seconds 0 --no-cpu-baseline --val-dice-steps 0 --no-secondary --serialize-streams > "$OUT/run.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY' > gpurun_out/pmc_lds.json
import csv, glob, json, re, sys, subprocess
sys.path.insert(0, ".")
import bench
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
    a = acc.setdefault(n, {"SQ_LDS_BANK_CONFLICT": 0.0, "SQ_LDS_IDX_ACTIVE": 0.0})
    if r["Counter_Name"] in a:
        a[r["Counter_Name"]] += float(r["Counter_Value"])
out = {"_meta": {"kernel_source_hash": bench.kernel_source_hash(), "definition": "conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, summed over the launches of one warm-up + one timed step"}}
for n, a in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_LDS_IDX_ACTIVE"]):
    if a["SQ_LDS_IDX_ACTIVE"] > 0:
        out[n] = {"lds_idx_active": a["SQ_LDS_IDX_ACTIVE"], "bank_conflict": a["SQ_LDS_BANK_CONFLICT"], "conflict_share": round(a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"], 4)}
print(json.dumps(out, indent=1))
PY
python3 -c "
import json; d=json.load(open('gpurun_out/pmc_lds.json'))
for k,v in list(d.items())[:14]: print(k[:70], v if k=='_meta' else v['conflict_share'])"
