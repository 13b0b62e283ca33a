# developer scratch: round-3 GPU session script (edited per experiment)
python -m pytest tests/test_gpu_bench_scale.py tests/test_gpu_neighbors.py tests/test_gpu_parity.py tests/test_gpu_tn2.py tests/test_gpu_et.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert|^E " | tail -12
