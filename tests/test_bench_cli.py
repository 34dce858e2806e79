"""Host-side checks of the benchmark command line (no GPU needed) and a short GPU run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "benchmark", "bench_blas3")
SUITES = os.path.join(ROOT, "benchmark", "config_csv")
sys.path.insert(0, os.path.join(ROOT, "benchmark"))


def _built():
    if not os.path.exists(BENCH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "benchmark")])


def _list(op, *extra):
    _built()
    out = subprocess.run([BENCH, op, "--benchmark_list_tests", *extra], check=True, capture_output=True, text=True)
    return out.stdout.split()


def test_gen_param_blocks():
    from gen_param import concat_ranges, nd_range, size_range, value_range

    # the two worked examples of the reference's generator (benchmark/README.md:197-232)
    flat = concat_ranges(nd_range(value_range(8), size_range(128, 512, 2)),
                         nd_range(size_range(128, 512, 2), value_range(8)))
    assert flat == [(8, 128), (8, 256), (8, 512), (128, 8), (256, 8), (512, 8)]
    gemv = nd_range(value_range("n", "t"), concat_ranges(value_range(32, 64), value_range(1024)),
                    size_range(128, 256, 2), value_range(1), value_range(0))
    assert len(gemv) == 12 and gemv[0] == ("n", 32, 128, 1, 0) and gemv[-1] == ("t", 1024, 256, 1, 0)


def test_csv_rows_become_benchmark_names(tmp_path):
    # the three example rows of benchmark/README.md:155-159, then the complex-scalar form :162-166
    csv = tmp_path / "p.csv"
    csv.write_text("n,n,42,42,42,1,0\nn,t,64,128,64,0.5,0.5\nt,n,13,3,7,0,0.7\n")
    assert _list("gemm", f"--csv-param={csv}") == ["BM_Gemm<float>/n/n/42/42/42/usm", "BM_Gemm<float>/n/t/64/128/64/usm",
                                                   "BM_Gemm<float>/t/n/13/3/7/usm"]
    csv.write_text("n,n,42,42,42,1,2,3,0\n")
    assert _list("gemm", f"--csv-param={csv}", "--dtype=double") == ["BM_Gemm<double>/n/n/42/42/42/usm"]
    csv.write_text("n,n,64,64,5000,1,0,32,interleaved\n")
    assert _list("gemm_batched", f"--csv-param={csv}") == ["BM_GemmBatched<float>/n/n/64/64/5000/32/interleaved/usm"]
    csv.write_text("l,u,n,n,345,560,1\n")
    assert _list("trsm", f"--csv-param={csv}") == ["BM_Trsm<float>/l/u/n/n/345/560/1/usm"]


def test_default_sweeps_match_the_reference_counts():
    # common/include/common/common_utils.hpp:469-490 (4 * 3^3), :725-744, :933-948, :1015-1032
    assert len(_list("gemm")) == 108
    assert len(_list("gemm_batched")) == 12
    assert len(_list("gemm_batched_strided")) == 12
    assert len(_list("symm")) == 20
    assert len(_list("trsm")) == 48


def test_bad_rows_are_rejected(tmp_path):
    _built()
    csv = tmp_path / "bad.csv"
    csv.write_text("n,n,42,42\n")
    res = subprocess.run([BENCH, "gemm", f"--csv-param={csv}", "--benchmark_list_tests"], capture_output=True, text=True)
    assert res.returncode == 2 and "invalid number of parameters (7 expected)" in res.stderr


def test_filter_and_suites():
    names = _list("gemm", f"--csv-param={SUITES}/gemm_square.csv", "--benchmark_filter=/(64|8192)/")
    assert names == ["BM_Gemm<float>/n/n/64/64/64/usm", "BM_Gemm<float>/n/n/8192/8192/8192/usm"]
    for op, name in [("gemm", "gemm_headline.csv"), ("gemm", "gemm_rectangular.csv"), ("gemm_batched", "gemm_batched.csv"),
                     ("gemm_batched_strided", "gemm_batched_strided.csv"), ("symm", "symm_square.csv"),
                     ("trsm", "trsm_square.csv")]:
        assert _list(op, f"--csv-param={SUITES}/{name}")


@pytest.mark.gpu
def test_benchmark_runs_on_the_gpu(tmp_path):
    _built()
    out = tmp_path / "r.json"
    for op, name, flt in [("gemm", "gemm_square.csv", "/(64|1024)/"), ("gemm_batched", "gemm_batched.csv", "/32/32/32/32/"),
                          ("symm", "symm_square.csv", "/256/"), ("trsm", "trsm_square.csv", "/l/l/n/n/512/")]:
        subprocess.run([BENCH, op, f"--csv-param={SUITES}/{name}", f"--benchmark_filter={flt}", "--benchmark_min_time=0.02",
                        "--benchmark_format=csv", f"--benchmark_out={out}", "--benchmark_out_format=json"],
                       check=True, capture_output=True, text=True, timeout=120)
        report = json.loads(out.read_text())
        assert report["benchmarks"], op
        for b in report["benchmarks"]:
            assert b["iterations"] >= 3 and b["avg_event_time"] > 0 and b["n_fl_ops"] > 0
    tune = subprocess.run([os.path.join(ROOT, "benchmark", "tune_gemm"), "300", "200", "100", "1", "3", "--trans=nt"],
                          check=True, capture_output=True, text=True, timeout=120)
    assert "auto" in tune.stdout and "RESULT DIFFERS" not in tune.stdout
