#!/bin/bash
# First GPU call for a branch that was written without a GPU (developer tool): check the branch out as a worktree under .wt_branch/,
# build its libraries HERE (hipcc cross-compiles), and run a command inside that tree on ONE box - after an optional command in the
# working tree (e.g. the full -m gpu suite of main), so both fit one gpurun call:
#   tools/validate_branch.sh wip/tn2-second-order 'python -m pytest tests/test_gpu_hvp.py -q -k tn2' 'python -m pytest tests -m gpu -x -q'
# Outputs the branch's run wrote to gpurun_out/ come back under gpurun_out/branch/.  DRY=1: build only, no GPU call.
set -e
BRANCH=$1
CMD=$2
MAIN_CMD=${3:-true}
TIMEOUT=${TIMEOUT:-1500}
ROOT=$(git rev-parse --show-toplevel)
cd "$ROOT"
git worktree add -q .wt_branch "$BRANCH"
trap 'cd "$ROOT"; rm -f gpurun_branch.sh; git worktree remove --force .wt_branch; git worktree prune' EXIT
(cd .wt_branch && python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|ok:" | tail -3)
cat > gpurun_branch.sh <<EOS
cd \$GRAFT_REPO_ROOT; echo "== working tree"; $MAIN_CMD
cd \$GRAFT_REPO_ROOT/.wt_branch; echo "== $BRANCH"; $CMD
mkdir -p \$GRAFT_REPO_ROOT/gpurun_out/branch; cp -r gpurun_out/. \$GRAFT_REPO_ROOT/gpurun_out/branch/ 2>/dev/null || true
EOS
if [ -n "$DRY" ]; then cat gpurun_branch.sh; exit 0; fi
/usr/local/graft/bin/gpurun --timeout $TIMEOUT -- 'bash gpurun_branch.sh' 2>&1 | tail -40
