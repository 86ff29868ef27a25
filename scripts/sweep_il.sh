mkdir -p gpurun_out/s31
for c in 2 3 4 6 8 16; do ATSC_IL_CH_PER_BLOCK=$c timeout -k 10 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s31/c$c.json 2>/dev/null || exit 1; done
