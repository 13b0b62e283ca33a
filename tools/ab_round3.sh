# developer scratch: round-3 GPU session script (edited per experiment)
python -m pytest tests/test_gpu_embed_rb.py tests/test_gpu_tn2.py tests/test_gpu_bench_scale.py -m gpu -q -x 2>&1 | tail -5
python bench.py --no-cpu-baseline --no-md --steps 10 --breakdown gpurun_out/bd_r3f.json > gpurun_out/bench_r3f.json 2> gpurun_out/bench_r3f.err
python -c "
import json
d=json.load(open('gpurun_out/bench_r3f.json'))
for k in ('water10k','tensornet2'):
    v=d.get(k,{}); print(k, v.get('ms_per_step'), v.get('classes_ms'), v.get('error'))
print('main', d['ms_per_step'])
"
