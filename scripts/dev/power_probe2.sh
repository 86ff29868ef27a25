#!/bin/bash
# GPU box: socket power and XCD clocks while the bench loops, per FIR form (read-only queries)
for form in wide fused; do
  python bench.py --steps 3000 --warmup 10 --no-cpu-baseline --no-extras --fir-form $form > gpurun_out/power_$form.json 2>/dev/null &
  BP=$!
  sleep 6
  echo "== $form"
  (/opt/rocm/bin/amd-smi metric -g 0 -p -c 2>&1 | grep -iE "socket_power|clk:|clk_locked|gfx_[0-7]|mem_0" -A1 | grep -iE "socket_power|clk: [0-9]" | head -14) || true
  wait $BP
  python -c "import json;d=json.load(open('gpurun_out/power_$form.json'));print(d['ms_per_step'], d['stage_ms']['mix_fir'])"
done
