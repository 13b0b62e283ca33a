# developer scratch: round-3 GPU session script (edited per experiment)
python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/pytest_r3e.log 2>&1; tail -6 gpurun_out/pytest_r3e.log
for sc in weak strong; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --rank-step graph --scaling $sc --no-cpu-baseline --no-md --no-aux > gpurun_out/bench_dist_$sc.json 2> gpurun_out/bench_dist_$sc.err
python -c "
import json
try:
    d=json.loads(open('gpurun_out/bench_dist_$sc.json').read().strip().splitlines()[-1])
    print('$sc', d['ms_per_step'], d.get('ranks_seen_by_rccl'), d.get('rccl'), d['config']['rank_step'][:60], d['roofline']['frac'], d['roofline'].get('measured','')[:40])
except Exception as e:
    print('$sc FAILED', e); print(open('gpurun_out/bench_dist_$sc.err').read()[-1500:])
"
done
python bench.py --no-cpu-baseline --breakdown gpurun_out/bd_r3e.json > gpurun_out/bench_r3e.json 2> gpurun_out/bench_r3e.err
python -c "
import json
d=json.load(open('gpurun_out/bench_r3e.json'))
for k in ('md_single_system','et_c4','et_c4_bf16','water10k','tensornet2'):
    v=d.get(k,{}); print(k, v.get('ms_per_step'), v.get('classes_ms'), v.get('error'))
print('main', d['ms_per_step'], d['roofline']['kernel'][:60], d['roofline']['frac'])
"
