#!/bin/bash
# Counter passes for every kernel family, one short process per (family, pass): tools/bin/one_family launches one
# shape 12 times with the family pinned (or left to the planner where a family cannot be pinned).
# usage: tools/pmc_families.sh      -> gpurun_out/pmc_families/<dtype>_<label>/{sq,tcc,fetch}_counter_collection.csv
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_families
mkdir -p "$O"
export TMPDIR=/tmp
cd /tmp
# label            M     N     K       batch kernel param
CASES="
huge_8192          8192  8192  8192    1     5 0
huge_4096          4096  4096  4096    1     5 0
large128_4096      4096  4096  4096    1     1 0
small64_4096       4096  4096  4096    1     2 0
small64_1024       1024  1024  1024    1     2 0
medium96_1536      1536  1536  1536    1     9 0
medium192_3072     3072  3072  3072    1     9 1
halves64_1024      1024  1024  2048    1     3 16842754
halves128_2048     2048  2048  4096    1     3 16908290
tiny32_batch       32    32    64      16384 7 0
stream_64cubed     64    64    64      4096  0 0
tall_skinny        256   256   1048576 1     0 0
naive_8x8_batch    8     8     8       65536 4 0
"
for dt in f32 f64; do
  echo "$CASES" | while read label m n k bs kern param; do
    [ -z "$label" ] && continue
    D=$O/${dt}_$label
    mkdir -p "$D"
    pass() {
      name=$1; shift
      timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d "$D" -o "$name" -- "$R/tools/bin/one_family" $dt $m $n $k $bs $kern $param > "$D/$name.log" 2>&1 || { echo "pass $name of $dt $label failed"; tail -3 "$D/$name.log"; return 1; }
    }
    pass sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE && \
    pass tcc TCC_HIT_sum TCC_MISS_sum && pass fetch FETCH_SIZE || exit 1
    echo "$dt $label: $(tail -1 "$D/sq.log")"
  done || exit 1
done
