# kernel table of the parity step at a given images/step (eager launches under the library profiler): bash tools/batch_table.sh 8
B=${1:-8}
python bench.py --batch $B --steps 20 --warmup 5 --no-side --no-parity --no-parity-at-batch --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bt_$B.json
python - <<PY
import json
d=json.load(open('gpurun_out/bt_$B.json'))
print("b=$B", round(d['value'],1), "img/s", round(d['ms_per_step'],3), "ms/step (graph replay)")
tot=0
for k in d['roofline'].get('kernels', []):
    tot+=k['ms_per_step']
    print(f"{k['name']:36s} {k['ms_per_step']:.3f} x{k['launches_per_step']:.0f} exec {k.get('executed_tflops') and round(k.get('executed_tflops'))} gbs {k.get('gbs') and round(k.get('gbs'))}")
print("sum of top 16:", round(tot,3))
PY
