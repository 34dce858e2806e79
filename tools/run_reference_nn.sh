#!/bin/bash
# The reference's AlexNet / VGG / ResNet GEMM shape suites (benchmark/config_csv/reference_nn/, the rows of the
# reference's benchmark/config_csv/blas3/gemm/gemm_{inference,training}_*.csv) through bench_blas3 for float and
# bf16, and every unique shape through the family tuner (float) for the planner-vs-best-candidate ratios.
# SUITE=reference_lm runs the reference's language-model suites (BERT, GPT-J, transformer:
# benchmark/config_csv/reference_lm/, the rows of benchmark/config_csv/blas3/gemm/language_models/*.csv) instead.
# usage: [SUITE=reference_nn|reference_lm] [SKIP_TUNER=1 | ONLY_TUNER=1] tools/run_reference_nn.sh    -> gpurun_out/$SUITE/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
SUITE=${SUITE:-reference_nn}
out=$R/gpurun_out/$SUITE
mkdir -p $out
cd $R
for f in benchmark/config_csv/$SUITE/*.csv; do
  [ -n "${ONLY_TUNER:-}" ] && break
  tag=$(basename $f .csv)
  for dt in float bf16; do
    timeout -k 10 170 benchmark/bench_blas3 gemm --csv-param=$f --dtype=$dt --benchmark_min_time=0.05 --benchmark_format=console \
      > $out/${tag}_$dt.txt 2> $out/${tag}_$dt.err || { echo "bench $tag $dt failed"; tail -3 $out/${tag}_$dt.err; exit 1; }
  done
  echo "suite $tag done"
done
[ -n "${SKIP_TUNER:-}" ] && exit 0
rm -f $out/tune_f32.csv $out/tune_f32.log
cat benchmark/config_csv/$SUITE/*.csv | grep -v "^#" | sort -u | while IFS=, read ta tb m k n al be; do
  timeout -k 10 60 benchmark/tune_gemm $m $k $n 1 20 --trans=nn --dtype=float --csv=$out/tune_f32.csv >> $out/tune_f32.log 2>&1 || { echo "tune failed $m $k $n"; exit 1; }
done
echo "tuner done"
python3 tools/plan_regret.py $out/tune_f32.csv --list 1.06 > $out/planner_vs_best_f32.txt
head -5 $out/planner_vs_best_f32.txt
