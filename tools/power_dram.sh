#!/bin/bash
# rocm-smi samples beside tools/probes/probe_dram_locality (two read patterns over the same 40 GB, ~8 s each)
tools/probes/build/probe_dram_locality 8 &
PID=$!
for i in $(seq 1 34); do
  echo "[t=$i] $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Current Socket Graphics Package Power|sclk clock level' | sed -E 's/^GPU\[0\]\s*:\s*//' | tr '\n' ';')"
  sleep 0.6
done
wait $PID
