#!/bin/bash
# Slab-length and slab-count probes of the K cut through the cube on the 64 x 64 tile (tools/bin/one_family, pinned cuts,
# best of five loops of 200 calls): time against slab length at a fixed slab count, and the same cut on aligned and
# unaligned shapes.  Outputs kept as profiles/r03/suites/chunk_probe.txt and split_depth_probe.txt.
# usage (GPU box, repo root): tools/probe_slabs.sh
export ONE_FAMILY_TIME=1
run() { echo -n "$*: "; timeout -k 5 60 tools/bin/one_family f32 $1 $2 $3 1 3 $(( $4 | (1<<16) )) || exit 1; }
# 64x64, K and depth chosen for the slab length (in K steps of 32): K = steps*32*depth
for steps in 2 3 4 5 6 7 8; do run 64 64 $((steps*32*32)) 32; done
for steps in 2 3 4 5 6 7 8; do run 64 64 $((steps*32*43)) 43; done
for steps in 2 3 4 5 6; do run 256 256 $((steps*32*11)) 11; done
for steps in 2 3 4 5 6; do run 256 256 $((steps*32*16)) 16; done
run() { echo -n "$*: "; timeout -k 5 60 tools/bin/one_family $1 $2 $3 $4 1 3 $(( $5 | (1<<16) )) || exit 1; }
for dt in f32 f64; do
for shape in "256 256 1024" "248 248 1000" "256 256 1000" "248 248 1024" "264 264 1024"; do
  for d in 8 12 16; do run $dt $shape $d; done
done; done
