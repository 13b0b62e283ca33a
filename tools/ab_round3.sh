# developer scratch: round-3 GPU session script (edited per experiment)
python -m pytest tests -m gpu -q --maxfail=8 > gpurun_out/pytest_r3h.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_r3h.log | tail -3; grep -E "^FAILED|^E " gpurun_out/pytest_r3h.log | head -20
