#!/bin/bash
# Sweep of the family tuner over a grid of shapes; CSV lines go to gpurun_out/tune_sweep_<dtype>.csv.
dtype=${1:-float}
out=gpurun_out/tune_sweep_$dtype.csv
log=gpurun_out/tune_sweep_$dtype.log
rm -f $out
for m in 64 256 1024 2048; do
  for n in 64 256 1024 2048; do
    for k in 256 1024 4096 16384; do
      timeout -k 10 60 benchmark/tune_gemm $m $k $n 1 10 --trans=nn --dtype=$dtype --csv=$out >> $log 2>&1 || { echo "failed $m $k $n"; exit 1; }
    done
  done
  echo "m=$m done"
done
# (m k n) odd shapes, big squares, and the band products of the multi-GPU schedule (8192 rows, a few hundred columns)
for shape in "512 512 512" "1000 1000 1000" "4096 4096 4096" "49 4608 512" "784 1152 128" "32 4096 4096" "1 4096 4096" "512 3072 768" "3136 576 64" "3000 3000 3000" "1536 1536 1536" \
             "768 768 768" "3072 3072 3072" "6144 6144 6144" "8192 8192 8192" "8192 8192 128" "8192 8192 256" "8192 8192 512" "8192 8192 1024" "128 8192 8192" "512 4096 4096" "4100 4100 4100" \
             "1536 256 1536" "1536 4096 1536" "768 4096 768" "1152 2048 1152" "2304 2304 2304" "1440 512 1536" "960 8192 960" "1344 1344 1344" "2048 2048 2048" "4608 4608 4608" "384 16384 384" "576 8192 768"; do
  timeout -k 10 60 benchmark/tune_gemm $shape 1 10 --trans=nn --dtype=$dtype --csv=$out >> $log 2>&1 || { echo "failed $shape"; exit 1; }
done
echo sweep done
