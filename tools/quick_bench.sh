# quick headline check: parity b32, no sides / oracle / cpu baseline
python bench.py --steps 20 --warmup 5 --no-side --no-parity --no-parity-at-batch --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/qb.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/qb.json'))
print(d['value'], d['ms_per_step'])
for k in d['roofline'].get('kernels', []): print(f"{k['name']:36s} {k['ms_per_step']:.3f} {k.get('executed_tflops')}")
PY
