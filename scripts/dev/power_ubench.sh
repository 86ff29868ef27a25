#!/bin/bash
# GPU box: power and clocks under the three loads of scripts/ubench/power_mix.hip (read-only queries)
for mode in 0 1 2; do
  scripts/ubench/bin/power_mix $mode 5 > gpurun_out/power_mix_$mode.txt &
  BP=$!
  sleep 2.5
  /opt/rocm/bin/amd-smi metric -g 0 -p -c 2>/dev/null | grep -E "SOCKET_POWER|GFX_[0-7]:|^ *CLK:" | tr -s ' ' | tr '\n' ' ' | sed 's/GFX_\([0-7]\): CLK: /x\1=/g' | cut -c1-220
  echo
  wait $BP || exit 1
  cat gpurun_out/power_mix_$mode.txt
done
