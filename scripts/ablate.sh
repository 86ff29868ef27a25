mkdir -p gpurun_out/s10
for a in 5 13 4 12; do
  ATSC_HIP_LIB=$PWD/build/ab$a/libatsc_hip.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s10/ab$a.json 2> gpurun_out/s10/ab$a.err || exit 1
done
