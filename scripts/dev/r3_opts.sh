#!/bin/bash
# GPU box: every A/B configuration of bench.py still runs (short)
set -o pipefail
O=gpurun_out/opts; mkdir -p $O
i=0
for o in "--iq-format cs16" "--upsample 2" "--fir-form fft" "--fir-form wide" "--symbol-tap" "--resample 8/7 --iq-format cs16" "--mode timesplit" "--mode timesplit --channels 64 --fields 16 --rccl"; do
  i=$((i+1))
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 2 $o > $O/opt$i.json 2> $O/opt$i.err || { echo "FAILED $o"; tail -5 $O/opt$i.err; exit 1; }
  python - "$o" $O/opt$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["ms_per_step"], (d.get("stage_ms") or {}).get("mix_fir"), (d.get("roofline") or {}).get("frac"), len(json.dumps(d)))
PY
done
