#!/bin/bash
# developer helper: A/B two builds on ONE GPU box (box-to-box scatter of the bench is +-2 %, more than most kernel changes).
# Usage (build container): tools/ab_build.sh   -> lib/libtmdnet_amd_new.so (working tree) and lib/libtmdnet_amd_old.so (HEAD);
# then on the box:  for v in old new old new; do cp lib/libtmdnet_amd_$v.so lib/libtmdnet_amd.so; python bench.py ...; done
set -e
cd "$(dirname "$0")/.."
L=torchmd-net_amd/lib
python __graft_entry__.py > /dev/null
cp $L/libtmdnet_amd.so $L/libtmdnet_amd_new.so
git stash -q
python __graft_entry__.py > /dev/null
cp $L/libtmdnet_amd.so $L/libtmdnet_amd_old.so
git stash pop -q
python __graft_entry__.py > /dev/null
echo "built: $L/libtmdnet_amd_{old,new}.so (lib/libtmdnet_amd.so = working tree)"
