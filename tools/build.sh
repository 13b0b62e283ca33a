#!/bin/bash
# build the HIP library from the repo root; non-zero exit on any compiler error (use with && before a gpurun)
cd "$(dirname "$0")/.." && python -c "import __graft_entry__ as g; g.build_hip()" > /tmp/build.log 2>&1
rc=$?
grep -E " error|warning:" /tmp/build.log | head -8
exit $rc
