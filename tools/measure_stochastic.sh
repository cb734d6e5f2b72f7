#!/bin/bash
# Repeats the tests whose bars are trajectory properties (not parity errors) under FMRI_MEASURE=1 and keeps each run's values, so that
# their limits can be set from the measured spread (every bar() comes from repeated measurements).  Usage: tools/measure_stochastic.sh [runs]
runs=${1:-5}
mkdir -p gpurun_out/stochastic
for i in $(seq 1 $runs); do
    rm -f gpurun_out/bars_measured.json
    FMRI_MEASURE=1 python -m pytest -q -m gpu \
        "tests/test_gpu_engine.py::test_training_loss_decreases_bf16" \
        "tests/test_gpu_adversarial.py::test_discriminator_model_learns_and_round_trips" > gpurun_out/stochastic/run_$i.log 2>&1
    cp gpurun_out/bars_measured.json gpurun_out/stochastic/bars_$i.json
done
python - <<'PY'
import glob, json
agg = {}
for f in sorted(glob.glob("gpurun_out/stochastic/bars_*.json")):
    for k, v in json.load(open(f)).items():
        agg.setdefault(k, {"limit": v["limit"], "runs": []})["runs"].append(v["measured"])
json.dump(agg, open("gpurun_out/stochastic/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(agg, indent=1, sort_keys=True))
PY
