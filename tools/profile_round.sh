#!/bin/bash
# Round profile recipe (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats, PMC HBM traffic.
# The snapshot has no .git: pass the commit the library was built from, e.g.  gpurun -- "GIT_COMMIT=$(git rev-parse --short HEAD) tools/profile_round.sh"
export GIT_COMMIT=${GIT_COMMIT:-unknown}
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
python $R/bench.py --breakdown $R/gpurun_out/breakdown.json > $R/gpurun_out/bench.json 2> $R/gpurun_out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-md --no-aux > $R/gpurun_out/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rd -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-md --no-aux > $R/gpurun_out/pmc_rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wr -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-md --no-aux > $R/gpurun_out/pmc_wr.log 2>&1
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_rd $R/gpurun_out/pmc_wr $R/gpurun_out/pmc_traffic.json
# the auxiliary legs (ET configs[3] in both storage modes, 10k-atom water box, TensorNet2): kernel stats + the same two PMC passes
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_aux -- python $R/bench.py --aux-only --steps 3 --warmup 2 > $R/gpurun_out/prof_aux.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rd_aux -- python $R/bench.py --aux-only --steps 1 --warmup 1 > $R/gpurun_out/pmc_rd_aux.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wr_aux -- python $R/bench.py --aux-only --steps 1 --warmup 1 > $R/gpurun_out/pmc_wr_aux.log 2>&1
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_rd_aux $R/gpurun_out/pmc_wr_aux $R/gpurun_out/pmc_traffic_aux.json "python bench.py --aux-only --steps 1 --warmup 1"
find $R/gpurun_out/pmc_rd_aux $R/gpurun_out/pmc_wr_aux -name '*.csv' -size +4M -delete
find $R/gpurun_out/prof_aux -name '*kernel_trace.csv' -size +8M -delete
# one rank's step of the halo exchange (98k-atom box, 8 slabs: bench.py leg one_system_8_slabs): kernel stats + the two PMC passes
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_halo -- python $R/tools/halo_profile.py > $R/gpurun_out/prof_halo.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_rd_halo -- python $R/tools/halo_profile.py > $R/gpurun_out/pmc_rd_halo.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wr_halo -- python $R/tools/halo_profile.py > $R/gpurun_out/pmc_wr_halo.log 2>&1
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_rd_halo $R/gpurun_out/pmc_wr_halo $R/gpurun_out/pmc_traffic_halo.json "python tools/halo_profile.py"
find $R/gpurun_out/pmc_rd_halo $R/gpurun_out/pmc_wr_halo -name '*.csv' -size +4M -delete
# issue-side counters of the dominant kernels (the request ceiling bench.py prints next to the HBM one): six --pmc passes
python $R/tools/pmc_issue.py $R/gpurun_out/pmc_issue.json > $R/gpurun_out/pmc_issue.log 2>&1
# keep only the summaries (counter CSVs of every dispatch are large)
find $R/gpurun_out/pmc_rd $R/gpurun_out/pmc_wr -name '*.csv' -size +4M -delete
find $R/gpurun_out/prof -name '*kernel_trace.csv' -size +8M -delete
cat $R/gpurun_out/bench.json
