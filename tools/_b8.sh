python bench.py --batch 8 --steps 40 --warmup 8 --no-side --no-parity --no-parity-at-batch --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/qb8.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/qb8.json'))
print(d['value'], d['ms_per_step'], d['median_ms'], d['p10_ms'], d['p90_ms'])
print(d.get('per_step_ms') or d.get('config'))
PY
