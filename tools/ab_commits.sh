#!/bin/bash
# Same-box A/B of two commits (developer tool).  Box-to-box scatter on the pool is +-3 %, which is more than most kernel changes
# are worth: build both libraries HERE (hipcc cross-compiles), then run the same command alternately in both trees on ONE box:
#   tools/ab_commits.sh <old-commit> 'python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-md --no-aux | tail -1 | cut -c1-160'
# The old commit is checked out as a git worktree under .ab_old/ (removed afterwards); the new side is the working tree.
set -e
OLD=$1; shift
CMD="$*"
ROOT=$(git rev-parse --show-toplevel)
cd "$ROOT"
git worktree add -q .ab_old "$OLD"
(cd .ab_old && python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|ok:" | tail -1)
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -E "error|ok:" | tail -1
cat > gpurun_ab.sh <<EOS
for r in 1 2; do
  for side in .ab_old .; do
    cd \$GRAFT_REPO_ROOT/\$side; echo "== \$side (round \$r)"; $CMD
  done
done
EOS
/usr/local/graft/bin/gpurun --timeout 900 -- 'bash gpurun_ab.sh' 2>&1 | tail -20
rm -f gpurun_ab.sh
git worktree remove --force .ab_old
git worktree prune
