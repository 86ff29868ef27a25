# usage: bash scripts/ablate.sh <outdir> <ablation ids...>   (GPU box; libs prebuilt into build/ab<id>/)
out=$1; shift
mkdir -p gpurun_out/$out
for a in "$@"; do
  ATSC_HIP_LIB=$PWD/build/ab$a/libatsc_hip.so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$out/ab$a.json 2> gpurun_out/$out/ab$a.err || exit 1
done
