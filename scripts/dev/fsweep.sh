for F in 64 63 65 60 62 64; do timeout -k 10 120 python bench.py --fields $F --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms']; print($F, 'step', d['ms_per_step'], 'fir', s['mix_fir'], 'fir/field us', round(s['mix_fir']/$F*1000,3), 'gsym', d['value'])" || exit 1; done
