#!/bin/bash
# Row b' (VERDICT r04): run the REFERENCE's own test files, unmodified, against torchmdnet_amd under the `torchmdnet` alias on the GPU.
# Step 1 (build container, /root/reference present): stage byte-identical copies of the reference's test files, the example YAMLs
# they read and its calculators.py into .ref_stage/ (git-ignored: reference sources never enter the history; NOT gpurun-ignored:
# the directory travels to the GPU box like the built .so).  The two checkpoints the reference's tests load are absent from the
# reference checkout itself (.MISSING_LARGE_BLOBS): stand-ins in the reference's checkpoint format are written by
# tools/ref_harness/make_example_ckpts.py (random-init, seed 1234).
# Step 2 (GPU box):  cd .ref_stage/tests && python -m pytest -q -p no:cacheprovider <files>   ->  gpurun_out/reference_tests.json
# usage: tools/run_reference_tests.sh stage | run | gpurun
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REF=${TMDNET_REFERENCE_ROOT:-/root/reference}
FILES="test_model.py test_calculator.py test_neighbors.py test_staticshapes.py test_equivariance.py test_examples.py"
stage() {
  rm -rf "$ROOT/.ref_stage"
  mkdir -p "$ROOT/.ref_stage/tests" "$ROOT/.ref_stage/examples" "$ROOT/.ref_stage/torchmdnet"
  for f in $FILES utils.py expected.pkl caffeine.pdb; do cp "$REF/tests/$f" "$ROOT/.ref_stage/tests/"; done
  cp "$REF"/examples/*.yaml "$ROOT/.ref_stage/examples/"
  cp "$REF/torchmdnet/calculators.py" "$ROOT/.ref_stage/torchmdnet/"
  cp "$ROOT/tools/ref_harness/conftest.py" "$ROOT/.ref_stage/tests/conftest.py"
  (cd "$ROOT/.ref_stage/tests" && sha256sum $FILES utils.py expected.pkl > ../staged.sha256)
  python "$ROOT/tools/ref_harness/make_example_ckpts.py" "$ROOT/.ref_stage/tests"
}
run() {
  cd "$ROOT/.ref_stage/tests"
  # one process per test file (a hard crash inside torch - e.g. capture_end with a captured backward on this image - must not take
  # the other files' results with it); test_neighbors.py under pytest-xdist, which reports a crashed worker as a failed test
  rm -f "$ROOT"/gpurun_out/reference_tests_part_*.json
  for f in $FILES; do
    extra=""; [ "$f" = test_neighbors.py ] && extra="-n 2"
    REF_TEST_REPORT="$ROOT/gpurun_out/reference_tests_part_${f%.py}.json" timeout ${REF_TEST_TIMEOUT:-1200} python -m pytest -q -p no:cacheprovider $extra $f 2>&1 | tail -3
  done
  python "$ROOT/tools/ref_harness/merge_reports.py" "$ROOT/gpurun_out" "$ROOT/gpurun_out/reference_tests.json"
}
case "$1" in
  stage) stage ;;
  run) run ;;
  gpurun) stage; trap 'rm -rf "$ROOT/.ref_stage"' EXIT   # the staged copies of reference files leave the tree again
          /usr/local/graft/bin/gpurun --timeout ${REF_GPURUN_TIMEOUT:-1800} -- 'bash tools/run_reference_tests.sh run' ;;
  *) echo "usage: $0 stage | run | gpurun"; exit 2 ;;
esac
