mkdir -p gpurun_out/s32
for c in 4 8 16 32 64; do ATSC_FIR_CH_PER_TASK=$c timeout -k 10 100 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s32/c$c.json 2>/dev/null || exit 1; done
