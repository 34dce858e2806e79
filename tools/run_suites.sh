#!/bin/bash
# Runs the benchmark suites and the tuner on a few shapes; writes reports under gpurun_out/suites/.
set -e
out=gpurun_out/suites
mkdir -p $out
B=benchmark/bench_blas3
S=benchmark/config_csv
run() { # op csv dtype tag
  timeout -k 10 170 $B $1 --csv-param=$S/$2 --dtype=$3 --benchmark_min_time=0.1 --benchmark_format=console \
    --benchmark_out=$out/$4.json --benchmark_out_format=json > $out/$4.txt 2>$out/$4.err
  echo "finished $4"
}
run gemm gemm_square.csv float gemm_square_f32
run gemm gemm_square.csv double gemm_square_f64
run gemm gemm_square.csv bf16 gemm_square_bf16
run gemm gemm_square.csv half gemm_square_f16
run gemm gemm_square_all_trans.csv float gemm_trans_f32
run gemm gemm_square_all_trans.csv double gemm_trans_f64
run gemm gemm_rectangular.csv float gemm_rect_f32
run gemm gemm_headline.csv float gemm_headline_f32
run gemm gemm_transformer_shapes.csv float gemm_transformer_f32
run gemm gemm_transformer_shapes.csv bf16 gemm_transformer_bf16
run gemm_batched gemm_batched_reference_shapes.csv float gemm_batched_reference_shapes_f32
run gemm_batched gemm_batched.csv float gemm_batched_f32
run gemm_batched_strided gemm_batched_strided.csv float gemm_batched_strided_f32
run symm symm_square.csv float symm_f32
run trsm trsm_square.csv float trsm_f32
run trsm trsm_square.csv double trsm_f64
for shape in "1024 1024 1024" "2048 2048 2048" "512 512 512" "3136 576 64" "256 65536 256" "49 4608 512" "4096 4096 4096"; do
  timeout -k 10 120 benchmark/tune_gemm $shape 1 5 --trans=nn >> $out/tune_f32.txt 2>&1
  echo "tuned $shape"
done
timeout -k 10 120 benchmark/tune_gemm 64 64 64 4096 5 --trans=nn >> $out/tune_f32.txt 2>&1
timeout -k 10 120 benchmark/tune_gemm 64 64 64 4096 5 interleaved --trans=nn >> $out/tune_f32.txt 2>&1
timeout -k 10 120 benchmark/tune_gemm 1024 1024 1024 1 5 --dtype=double --trans=nn >> $out/tune_f64.txt 2>&1
echo all done
