#!/bin/bash
# Four K quarters in one workgroup on the 32 x 32 tile against the planner's choice, fp64 and fp32 (the constants
# quarter_k / quarter_over of the cost model come from this table: profiles/r03/suites/quarters_probe.txt).
# usage (GPU box, repo root): tools/probe_quarters.sh
export ONE_FAMILY_TIME=1
Q=$(( 4 | (4<<16) | (1<<24) ))
for dt in f64 f32; do
for shape in "512 512 512" "256 256 1024" "384 384 768" "512 512 2048" "768 768 768" "512 256 512" "1024 1024 512" "640 640 640"; do
  echo -n "$dt $shape auto: "; timeout -k 5 60 tools/bin/one_family $dt $shape 1 0 0 || exit 1
  echo -n "$dt $shape quarters: "; timeout -k 5 60 tools/bin/one_family $dt $shape 1 3 $Q || exit 1
done; done
