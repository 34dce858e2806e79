#!/usr/bin/env python3
"""Generate a CSV parameter file for bench_blas3 from a range expression.

Same four building blocks as the reference's generator (benchmark/README.md:173-243):

  size_range(low, high, mult)   low, low*mult, ... up to and including high
  value_range(v1, ..., vn)      exactly these values
  concat_ranges(r1, ..., rn)    all rows of r1, then r2, ...
  nd_range(r1, ..., rn)         every combination, first range varying slowest

  gen_param.py -o gemm.csv -e "nd_range(value_range('n','t'), value_range('n','t'),
      size_range(128,1024,2), size_range(128,1024,2), size_range(128,1024,2), value_range(1), value_range(0))"
"""
import argparse
import itertools
import sys


def _rows(r):
    return [x if isinstance(x, tuple) else (x,) for x in r]


def size_range(low, high, mult):
    if low <= 0 or mult <= 1:
        raise ValueError("size_range needs low > 0 and mult > 1")
    out, v = [], low
    while v <= high:
        out.append((v,))
        v *= mult
    return out


def value_range(*values):
    return [(v,) for v in values]


def concat_ranges(*ranges):
    return [row for r in ranges for row in _rows(r)]


def nd_range(*ranges):
    return [tuple(itertools.chain.from_iterable(combo)) for combo in itertools.product(*map(_rows, ranges))]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-o", "--output", required=True, help="CSV file to write ('-' for stdout)")
    ap.add_argument("-e", "--expr", required=True, help="range expression")
    args = ap.parse_args(argv)
    scope = {"__builtins__": {}, "size_range": size_range, "value_range": value_range,
             "concat_ranges": concat_ranges, "nd_range": nd_range}
    rows = _rows(eval(args.expr, scope))  # the expression language is these four functions and literals
    text = "".join(",".join(str(c) for c in row) + "\n" for row in rows)
    if args.output == "-":
        sys.stdout.write(text)
    else:
        with open(args.output, "w") as f:
            f.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
