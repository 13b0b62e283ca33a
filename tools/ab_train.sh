#!/bin/bash
# same-box A/B of the training-step times with the 128 x 128 weight-gradient kernel on / off (developer tool)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1; do
  python tools/train_step_time.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['training_c2']; print('gemm128', round(t['ms_per_step_energy_only'],2), round(t['ms_per_step_energy_and_forces'],2), round(d['second_order_pass_ms'],2))"
  TMDNET_TN_GEMM64=1 python tools/train_step_time.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); t=d['training_c2']; print('gemm64 ', round(t['ms_per_step_energy_only'],2), round(t['ms_per_step_energy_and_forces'],2), round(d['second_order_pass_ms'],2))"
done
