#!/bin/bash
# GPU box: sample power and clocks while the bench loops (read-only queries)
python bench.py --steps 4000 --warmup 10 --no-cpu-baseline > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
sleep 5
/opt/rocm/bin/rocm-smi --showpower --showmaxpower --showclocks 2>&1 | grep -v "^=\|^$" | head -30
(/opt/rocm/bin/amd-smi metric -g 0 -p -c 2>&1 | head -60) || true
wait $BP
tail -c 200 gpurun_out/power_bench.json
