#!/bin/bash
# Band height of the workgroup -> tile map (SBAMD_GROUP_M) against time, fetched bytes and L2 hit rate for the
# 8192^3 headline kernels.  Each figure comes from its own run (timing un-profiled; one rocprofv3 --pmc pass each
# for FETCH_SIZE and for TCC_HIT_sum / TCC_MISS_sum).   -> gpurun_out/group_m/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/group_m
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for dt in f32 f64; do
  for gm in 2 4 8 16; do
    export SBAMD_GROUP_M=$gm
    timeout -k 10 120 python3 $R/bench.py --dtype $dt --no-cpu --no-extras > $O/${dt}_gm${gm}.json 2> $O/${dt}_gm${gm}.err || { echo "bench $dt $gm failed"; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${dt}_gm${gm}_fetch -o out -- python3 $R/bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu --no-extras > $O/${dt}_gm${gm}_fetch.log 2>&1 || { echo "fetch $dt $gm failed"; exit 1; }
    timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${dt}_gm${gm}_tcc -o out -- python3 $R/bench.py --dtype $dt --steps 3 --warmup 1 --no-cpu --no-extras > $O/${dt}_gm${gm}_tcc.log 2>&1 || { echo "tcc $dt $gm failed"; exit 1; }
    echo "$dt group_m $gm done"
  done
done
unset SBAMD_GROUP_M
