#!/bin/bash
# round 4, visit 25: what of a resident LSTM costs the other stream's GEMMs?  The per-layer kernel (no stack) with
# its timing probes: APS_LSTM_DEBUG=1 no gather (no hand-off traffic, no waiting; same occupancy and compute),
# 2 gather without waiting (traffic, no waiting), 0 the real thing
set -u
O=gpurun_out/r04_s25; mkdir -p $O
export APS_NO_LSTM_STACK=1
for dbg in 0 1 2 3; do
  echo "== per-layer LSTM kernels, APS_LSTM_DEBUG=$dbg" | tee -a $O/probe.txt
  APS_LSTM_DEBUG=$dbg timeout 300 python scripts/gemm_sequence_overlap.py 2>&1 | grep "LSTM stack on another" | tee -a $O/probe.txt
done
