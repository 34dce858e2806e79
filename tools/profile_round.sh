#!/bin/bash
# Round profile: bench JSON lines, rocprofv3 kernel-trace stats and the counter passes, for fp32 and fp64.
# usage: tools/profile_round.sh <round tag, e.g. r01>     (writes gpurun_out/profile_<tag>/)
set -u
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/profile_$tag
mkdir -p "$O"
export TMPDIR=/tmp
cd "$R"
timeout -k 10 400 python3 bench.py > "$O/bench_sgemm8192.json" 2> "$O/bench_sgemm8192.err" || exit 1
echo "bench f32 done"
timeout -k 10 400 python3 bench.py --dtype f64 > "$O/bench_dgemm8192.json" 2> "$O/bench_dgemm8192.err" || exit 1
echo "bench f64 done"
cd /tmp
for dt in f32 f64; do
  # the headline kernel alone (its average must agree with the bench line) ...
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$dt" -o trace -- python3 "$R/bench.py" --dtype $dt --no-cpu --no-extras > "$O/trace_$dt.log" 2>&1 || { echo "trace $dt failed"; tail -5 "$O/trace_$dt.log"; exit 1; }
  echo "trace $dt done"
done
# ... and every kernel of the side measurements (batched, tall-skinny, 16-bit)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_extras" -o trace -- python3 "$R/bench.py" --no-cpu > "$O/trace_extras.log" 2>&1 || { echo "trace extras failed"; exit 1; }
cd "$R"
tools/pmc_pass.sh ${tag}_f32nn && tools/pmc_pass.sh ${tag}_f64nn --dtype f64
