"""Generates tests/golden/gemm_golden.npz: small seeded GEMM cases (inputs and the
oracle's outputs) in every arithmetic form.  The reference holds no golden vectors
for GEMM and cannot be run here, so these fixtures pin the ORACLE against silent
drift (compiler flags, refactors); the oracle itself is pinned against OpenBLAS in
tests/test_oracle.py.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle  # noqa: E402

CASES = [
    # name, dtype, form, ta, tb, m, n, k, lda_mul, ldb_mul, ldc_mul, batch, alpha, beta, depth, k_chunk
    ("s_nn_ref", np.float32, oracle.FORM_REF, 0, 0, 11, 16, 17, 1, 1, 1, 1, 1.5, 1.5, 1, 0),
    ("s_tn_ref_ld", np.float32, oracle.FORM_REF, 1, 0, 32, 11, 17, 2, 3, 4, 1, 1.5, 0.0, 1, 0),
    ("s_nt_local", np.float32, oracle.FORM_LOCAL, 0, 1, 16, 16, 16, 1, 1, 1, 1, 1.5, 1.5, 1, 0),
    ("s_tt_batched", np.float32, oracle.FORM_REF, 1, 1, 13, 9, 21, 1, 1, 1, 3, 3.0, 7.0, 1, 0),
    ("s_nn_splitk", np.float32, oracle.FORM_SPLITK, 0, 0, 7, 9, 515, 1, 1, 1, 1, 1.5, 0.5, 5, 112),
    ("s_nn_splitk_tree", np.float32, oracle.FORM_SPLITK_TREE, 0, 0, 7, 9, 515, 1, 1, 1, 1, 1.5, 0.5, 5, 112),
    ("d_nn_ref", np.float64, oracle.FORM_REF, 0, 0, 11, 16, 17, 1, 1, 1, 1, 1.5, 1.5, 1, 0),
    ("d_nt_ref_ld", np.float64, oracle.FORM_REF, 0, 1, 32, 11, 17, 2, 3, 4, 1, 1.5, 0.0, 1, 0),
    ("d_tn_local", np.float64, oracle.FORM_LOCAL, 1, 0, 16, 16, 16, 1, 1, 1, 1, 1.5, 1.5, 1, 0),
    ("d_tt_splitk", np.float64, oracle.FORM_SPLITK, 1, 1, 7, 9, 515, 1, 1, 1, 1, 1.5, 0.5, 5, 112),
]


def main():
    out = {}
    for i, (name, dtype, form, ta, tb, m, n, k, lam, lbm, lcm, batch, alpha, beta, depth, k_chunk) in enumerate(CASES):
        rng = np.random.default_rng(12345 + i)  # 12345: the reference's nominal seed (test/unittest/main.cpp:60)
        lda = (k if ta else m) * lam
        ldb = (n if tb else k) * lbm
        ldc = m * lcm
        sa, sb, sc = m * k * lam, k * n * lbm, m * n * lcm
        a = rng.uniform(-2, 5, batch * sa).astype(dtype)
        b = rng.uniform(-2, 5, batch * sb).astype(dtype)
        c = rng.uniform(-2, 5, batch * sc).astype(dtype)
        res = c.copy()
        st = oracle.gemm_backend("nt"[ta], "nt"[tb], m, n, k, alpha, a, lda, sa, b, ldb, sb, beta, res, ldc, sc,
                                 batch=batch, form=form, depth=depth, k_chunk=k_chunk)
        assert st == 0
        out[name + ".meta"] = np.array([form, ta, tb, m, n, k, lda, ldb, ldc, batch, sa, sb, sc, depth, k_chunk], np.int64)
        out[name + ".scal"] = np.array([alpha, beta], np.float64)
        out[name + ".a"], out[name + ".b"], out[name + ".c"], out[name + ".out"] = a, b, c, res
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
