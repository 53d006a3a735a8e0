#!/bin/bash
# single-call detector latency of the batches under IIR-stage settings -> gpurun_out/sweep_hp_<kind>.log
set -e
mkdir -p gpurun_out
T='[{}, {"hp_dedupe": -1, "hp_early": -1}, {"hp_warm": 81920}, {"hp_warm": 122880}, {"hp_chunk": 16384}, {"hp_chunk": 16384, "hp_warm": 81920}, {"hp_chunk": 16384, "hp_warm": 122880}, {"hp_chunk": 8192, "hp_warm": 81920}, {"hp_chunk": 8192, "hp_warm": 122880}, {"hp_chunk": 32768, "hp_warm": 122880}, {"hp_candidates": 16, "hp_warm": 61440, "hp_chunk": 16384}, {"hp_candidates": 16, "hp_warm": 81920, "hp_chunk": 32768}]'
for k in "$@"; do
  timeout -k 10 500 python tools/tune_detect.py $k "$T" > gpurun_out/sweep_hp_$k.log 2>&1
  cat gpurun_out/sweep_hp_$k.log | grep tuning
done
