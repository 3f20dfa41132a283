#!/bin/bash
# GPU-box job (round 3): a long run of the random-program parity tests on the round's final kernels.  usage: tools/jobs/r3_fuzz.sh OUTDIR
out=$1; mkdir -p "$out"
RM_FUZZ_FIRST_SEED=${RM_FUZZ_FIRST_SEED:-3000} RM_FUZZ_SEEDS=400 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -p no:cacheprovider > "$out/fuzz.log" 2>&1; echo "fuzz rc=$?" > "$out/status.txt"
tail -3 "$out/fuzz.log"
RM_FUZZ1080_FIRST_SEED=${RM_FUZZ1080_FIRST_SEED:-100} RM_FUZZ1080_SEEDS=96 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz_1080p.py -x -q -m gpu -p no:cacheprovider > "$out/fuzz1080.log" 2>&1; echo "fuzz1080 rc=$?" >> "$out/status.txt"
tail -3 "$out/fuzz1080.log"
cat "$out/status.txt"
