#!/bin/bash
# tools/bf16_error_study.py once per DBM_BF16_FP32_LAYERS mask (read once per process); results -> gpurun_out/bf16_error_study.jsonl
mkdir -p gpurun_out
: > gpurun_out/bf16_error_study.jsonl
for m in ${MASKS:-0 1 16 17 19 25 27}; do
  DBM_BF16_FP32_LAYERS=$m python tools/bf16_error_study.py 2>/dev/null | tail -1 >> gpurun_out/bf16_error_study.jsonl
done
cat gpurun_out/bf16_error_study.jsonl
