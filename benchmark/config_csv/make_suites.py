#!/usr/bin/env python3
"""Writes the parameter suites kept next to this file (run from anywhere)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from gen_param import concat_ranges, nd_range, size_range, value_range  # noqa: E402


def write(name, rows):
    with open(os.path.join(HERE, name), "w") as f:
        for row in rows:
            f.write(",".join(str(c) for c in row) + "\n")


def cube(s):
    return nd_range(value_range(s), value_range(s), value_range(s))


squares = concat_ranges(*[cube(s) for s in (64, 128, 256, 512, 1024, 2048, 4096, 8192)])
nt = value_range("n", "t")
# transA,transB,m,k,n,alpha,beta
write("gemm_square.csv", nd_range(value_range("n"), value_range("n"), squares, value_range(1), value_range(0)))
write("gemm_square_all_trans.csv", nd_range(nt, nt, concat_ranges(cube(1024), cube(4096), cube(8192)),
                                            value_range(1), value_range(1)))
# the headline shapes: large squares, and the tall-skinny reduction shape (m, k, n order)
write("gemm_headline.csv", concat_ranges(
    nd_range(value_range("n"), nt, cube(8192), value_range(1), value_range(0)),
    nd_range(value_range("n"), value_range("n"), value_range(256), value_range(4194304), value_range(256),
             value_range(1), value_range(0))))
# rectangular shapes typical of im2col convolutions and transformer blocks (m, k, n)
rect = [(3136, 576, 64), (784, 1152, 128), (196, 2304, 256), (49, 4608, 512), (12544, 147, 64),
        (512, 768, 768), (512, 768, 3072), (512, 3072, 768), (2048, 4096, 4096), (2048, 4096, 16384),
        (32, 4096, 4096), (1, 4096, 4096), (4096, 32, 4096), (100, 100, 100), (1000, 1000, 1000)]
write("gemm_rectangular.csv", [(ta, tb, m, k, n, 1, 0) for (m, k, n) in rect for ta, tb in (("n", "n"), ("t", "n"))])
# transA,transB,m,k,n,alpha,beta,batch_size,batch_type
write("gemm_batched.csv", nd_range(value_range("n"), value_range("n"),
                                   concat_ranges(cube(16), cube(32), cube(64), cube(128), cube(256)),
                                   value_range(1), value_range(0), value_range(32, 1024),
                                   value_range("strided", "interleaved")))
# ...,batch_size,stride_a_mul,stride_b_mul,stride_c_mul
write("gemm_batched_strided.csv", nd_range(value_range("n"), nt, concat_ranges(cube(64), cube(256), cube(1024)),
                                           value_range(1), value_range(0), value_range(8, 256), value_range(1),
                                           value_range(1), value_range(1, 2)))
# side,uplo,m,n,alpha,beta
write("symm_square.csv", nd_range(value_range("l", "r"), value_range("l", "u"),
                                  concat_ranges(*[nd_range(value_range(s), value_range(s)) for s in (256, 1024, 4096)]),
                                  value_range(1), value_range(0)))
# side,uplo,trans,diag,m,n,alpha
write("trsm_square.csv", nd_range(value_range("l", "r"), value_range("l", "u"), nt, value_range("n"),
                                  concat_ranges(*[nd_range(value_range(s), value_range(s)) for s in (512, 2048, 8192)]),
                                  value_range(1)))
