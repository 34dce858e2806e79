"""The planner's choices for the shapes the design leans on, evaluated WITHOUT a GPU: the plan depends on the shape,
the element type and the compute-unit count only (tools/plan_probe.hip compiles plan_gemm for the host).  Guards the
cost model's constants against edits that would silently move a headline shape to another tile."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("needs hipcc")
    exe = str(tmp_path_factory.mktemp("plan") / "plan_probe")
    subprocess.run([HIPCC, "-O1", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{ROOT}/sycl_blas_amd/csrc",
                    os.path.join(ROOT, "tools", "plan_probe.hip"), "-o", exe], check=True, capture_output=True, timeout=300)

    def plan(dtype, m, n, k, cus=256):
        out = subprocess.run([exe, dtype, str(cus)], input=f"{m} {n} {k}\n", capture_output=True, text=True, check=True).stdout.split()
        return tuple(int(x) for x in out[3:])  # (tile rows, tile columns, K slabs, fused)

    return plan


def test_headline_shapes_take_the_biggest_tile_in_one_launch(probe):
    assert probe("float", 8192, 8192, 8192) == (256, 256, 1, 0)
    assert probe("double", 8192, 8192, 8192) == (256, 128, 1, 0)
    assert probe("float", 4096, 4096, 4096) == (256, 256, 1, 0)


def test_config1_is_one_launch_without_a_cut(probe):
    assert probe("float", 1024, 1024, 1024) == (64, 64, 1, 0)


def test_tall_skinny_cuts_k_on_a_big_tile(probe):
    tm, tn, depth, fused = probe("float", 256, 256, 4194304)
    assert (tm, tn) == (256, 256) and depth == 256 and not fused
    tm, tn, depth, fused = probe("float", 256, 256, 262144)
    assert tm >= 128 and depth >= 32 and not fused  # not sixteen lone 64 x 64 slabs (HBM latency bound)


def test_sizes_that_quantise_badly_take_the_96_and_192_tiles(probe):
    assert probe("float", 1536, 1536, 1536)[:2] == (96, 96)
    assert probe("float", 3072, 3072, 3072)[:2] in ((96, 96), (192, 192))
    assert probe("double", 3072, 3072, 3072)[:2] in ((96, 96), (192, 192))


def test_small_cubes_take_the_cuts_inside_one_workgroup(probe):
    assert probe("float", 512, 512, 512) == (32, 32, 4, 1)
    assert probe("double", 512, 512, 512) == (32, 32, 4, 1)
    assert probe("double", 1024, 1024, 1024) == (64, 64, 2, 1)


def test_vector_shaped_products_are_cut_deep(probe):
    tm, tn, depth, fused = probe("float", 1, 4096, 4096)
    assert depth >= 8 and not fused


def test_a_smaller_part_gets_plans_of_its_own(probe):
    # the model works in compute units: with a quarter of them 2048^3 already fills the chip with big tiles
    assert probe("float", 2048, 2048, 2048, cus=64)[:2] == (256, 256)
    assert probe("float", 2048, 2048, 2048, cus=256)[:2] != (256, 256)
