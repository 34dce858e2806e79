#!/bin/bash
# Everything the tracked evidence of a round is made from, in three GPU calls of at most 20 minutes each:
#   tools/refresh_evidence.sh profile [round tag]   bench lines, kernel traces, counter passes (headline + every family), both sweeps
#   tools/refresh_evidence.sh suites    the reference NN / LM suites through bench_blas3 and the tuner
#   tools/refresh_evidence.sh vendor    vendor comparison tables (four types) and the band probe
# Results land under gpurun_out/; tools/collect_profile.py <round> copies the profile set into profiles/<round>/.
set -u
case "${1:-profile}" in
profile)
  tools/profile_round.sh ${2:-r03} > gpurun_out/final_profile.log 2>&1 || { echo "profile failed"; tail -5 gpurun_out/final_profile.log; exit 1; }
  echo "profile done"
  rm -rf gpurun_out/pmc_families
  tools/pmc_families.sh > gpurun_out/pmc_families.log 2>&1 || { echo "families failed"; tail -5 gpurun_out/pmc_families.log; exit 1; }
  echo "families done"
  timeout -k 10 500 tools/tune_sweep.sh float 2>&1 | tail -1 || exit 1
  timeout -k 10 600 tools/tune_sweep.sh double 2>&1 | tail -1 || exit 1
  ;;
suites)
  SUITE=reference_lm tools/run_reference_nn.sh || exit 1
  SUITE=reference_nn tools/run_reference_nn.sh || exit 1
  ;;
vendor)
  mkdir -p gpurun_out/vendor
  for dt in f32 f64 bf16 f16; do
    timeout -k 10 280 python3 tools/compare_vendor.py $dt > gpurun_out/vendor/compare_vendor_$dt.txt 2> gpurun_out/vendor/compare_vendor_$dt.err || { echo "compare $dt failed"; tail -5 gpurun_out/vendor/compare_vendor_$dt.err; exit 1; }
    echo "compare $dt done"
  done
  for dt in f32 f64; do
    timeout -k 10 200 python3 tools/probe_bands.py $dt > gpurun_out/vendor/band_gemms_$dt.txt 2>&1 || { echo "bands $dt failed"; exit 1; }
    echo "bands $dt done"
  done
  ;;
esac
