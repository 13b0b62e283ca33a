# developer scratch: round-3 GPU session script (edited per experiment)
python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/pytest_r3c.log 2>&1; tail -15 gpurun_out/pytest_r3c.log
cat gpurun_out/et_bf16_error.json; echo
python bench.py --no-cpu-baseline --no-md --breakdown gpurun_out/bd_r3c.json > gpurun_out/bench_r3c.json 2> gpurun_out/bench_r3c.err
python -c "
import json
d=json.load(open('gpurun_out/bench_r3c.json'))
for k in ('et_c4','et_c4_bf16','water10k','tensornet2'):
    v=d.get(k,{}); print(k, v.get('ms_per_step'), v.get('classes_ms'), v.get('error'))
print('main', d['ms_per_step'])
"
