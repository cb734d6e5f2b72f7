#!/bin/bash
# Every measured configuration of DESIGN.md in one go (run ON the GPU box, from the repo root: gpurun -- 'bash tools/bench_all.sh').
# One JSON line per workload on stdout; the same lines land in gpurun_out/bench_all.jsonl.
set -u
ROOT=$PWD
OUT=$ROOT/gpurun_out/bench_all.jsonl
mkdir -p "$ROOT/gpurun_out"
: > "$OUT"
run() { echo "# $*" >&2; "$@" 2>/dev/null | tail -1 | tee -a "$OUT"; }
run python3 bench.py --no-cpu-baseline --val-dice-steps 0 --no-secondary                 # configs[1]: headline 3-D training step (two streams) + roofline objects
run python3 bench.py --no-cpu-baseline --val-dice-steps 0 --no-secondary --serialize-streams --no-launch-timing
run python3 tools/bench_2d.py                          # configs[3]: 2-D training step
run python3 tools/infer_volume.py                      # configs[4]: sliding-window inference of a 160x256x256 volume
run python3 tools/infer_volume_2d.py                   # 2-D sliding-window inference
run python3 tools/bench_isensee.py                     # isensee2017_model_3d defaults
run python3 tools/bench_variants.py                    # BatchNorm / InstanceNorm / Deconvolution3D variants of unet_model_3d
run python3 tools/bench_fit.py                         # through train_model -> fit_generator (host generator, device generator)
run python3 tools/bench_sampler.py                     # device patch sampler + augmentation
run python3 tools/bench_adversarial.py                # adversarial round: discriminator step + generator step through the frozen discriminator
