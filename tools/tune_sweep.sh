#!/bin/bash
# Sweep of the family tuner over a grid of shapes; CSV lines go to gpurun_out/tune_sweep.csv.
out=gpurun_out/tune_sweep.csv
log=gpurun_out/tune_sweep.log
dtype=${1:-float}
rm -f $out
for m in 64 256 1024 2048; do
  for n in 64 256 1024 2048; do
    for k in 256 1024 4096 16384; do
      timeout -k 10 60 benchmark/tune_gemm $m $k $n 1 10 --trans=nn --dtype=$dtype --csv=$out >> $log 2>&1 || { echo "failed $m $k $n"; exit 1; }
    done
  done
  echo "m=$m done"
done
for shape in "512 512 512" "1000 1000 1000" "4096 4096 4096" "49 4608 512" "784 1152 128" "32 4096 4096" "1 4096 4096" "512 3072 768" "3136 576 64" "3000 3000 3000" "1536 1536 1536"; do
  timeout -k 10 60 benchmark/tune_gemm $shape 1 10 --trans=nn --dtype=$dtype --csv=$out >> $log 2>&1 || { echo "failed $shape"; exit 1; }
done
echo sweep done
