#!/usr/bin/env python3
"""Writes tests/golden/reference_kat.json: the known-answer vectors the
reference's own hot-path tests hold, as DATA (inputs + expected outcome).

Source of every case (inputs are the inline literals / sin(i) loops there):
  /root/reference/tests/test_cross_correlation.c:21-113   (8 cases)
  /root/reference/tests/test_pearson_coefficient.c:20-58  (4 cases)

Run once in the build container:  python tests/golden/make_reference_kat.py
Nothing here needs /root/reference at test time.
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))

xc = []


def xcase(name, cite, source, sample, **expect):
    assert len(source) == 2 * len(sample)
    xc.append({"name": name, "cite": cite, "source": source, "sample": sample, "expect": expect})


# test_cross_correlation.c:21-29
xcase("xc1_identical", "tests/test_cross_correlation.c:21-29",
      [1.1, 2.2, 3.3, 4.4, 5.5, 0, 0, 0, 0, 0], [1.1, 2.2, 3.3, 4.4, 5.5],
      ret=0, lag=0, coef_eq=1.0)
# :31-38
xcase("xc2_zero_sample", "tests/test_cross_correlation.c:31-38",
      [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14], [0, 0, 0, 0, 0, 0, 0], ret=-1)
# :40-49
xcase("xc3_right", "tests/test_cross_correlation.c:40-49",
      [0, 0, 0, 1, 2, 3, 4, 5, 6, 0, 0, 0], [1, 2, 3, 4, 5, 6], ret=0, lag=3, coef_gt=0.95)
# :51-60
xcase("xc4_left", "tests/test_cross_correlation.c:51-60",
      [1, 2, 3, 0.4, 1.1, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 1, 2, 3], ret=0, lag=-3, coef_gt=0.95)
# :62-71
xcase("xc5", "tests/test_cross_correlation.c:62-71",
      [1, 2, 3, 4, -1.0, 0, 0, 4, 3, 2, 1, 0, 0, 0], [0, 0, 0, 1, 2, 3, 4],
      ret=0, lag=-3, coef_gt=0.95)
# :73-81
xcase("xc6", "tests/test_cross_correlation.c:73-81",
      [0, 0, 0, 0, 0, 1, 2, 3, 4, -1, -3, -5, 0, 0], [1, 2, 3, 4, -1, -3, -5],
      ret=0, lag=5, coef_gt=0.95)
# :83-96  sin(i), i as size_t -> double
n = 1000
xcase("xc7_sine", "tests/test_cross_correlation.c:83-96",
      [math.sin(float(i)) for i in range(2 * n)], [math.sin(float(i)) for i in range(n)],
      ret=0, lag=0, coef_gt=0.95)
# :98-113
xcase("xc8_neg_sine", "tests/test_cross_correlation.c:98-113",
      [math.sin(float(i + 180)) for i in range(n)] + [0.0] * n,
      [math.sin(float(i)) for i in range(n)], ret=0, lag=-1, coef_lt=-0.95)

pc = []


def pcase(name, cite, source_seg, sample_seg, **expect):
    assert len(source_seg) == len(sample_seg)
    pc.append({"name": name, "cite": cite, "source_seg": source_seg, "sample_seg": sample_seg,
               "expect": expect})


# test_pearson_coefficient.c:20-28 : lag = -2 -> source[0 .. len-2), sample[2 .. len)
s1 = [1.0, 2.1, 3.2, 4.3, 5.4, 6.5, 7.6, 8.7, 9.8, 10.9]
t1 = [0, 0, 1.0, 2.1, 3.2, 4.3, 5.4, 6.5, 7.6, 8.7]
pcase("pc1_left", "tests/test_pearson_coefficient.c:20-28", s1[0:8], t1[2:10], eq=1.0)
# :30-39 : lag = 4 -> source[4 .. 9), sample[0 .. 5)
s2 = [0, 0, 0, 0, 100, 200, 300, 400, 500, 600, 700]
t2 = [100, 200, 300, 400, 500]
pcase("pc2_right", "tests/test_pearson_coefficient.c:30-39", s2[4:9], t2, eq=1.0)
# :41-48
pcase("pc3_negative", "tests/test_pearson_coefficient.c:41-48", [1, 2, 3, 4], [4, 3, 2, 1], eq=-1.0)
# :50-56
pcase("pc4_nan", "tests/test_pearson_coefficient.c:50-56", [1, 2, 3, 4], [0, 0, 0, 0], nan=True)

with open(os.path.join(HERE, "reference_kat.json"), "w") as f:
    json.dump({"cross_correlation": xc, "pearson_coefficient": pc}, f)

# plain-text copy for tests/c/kat_runner.c (no JSON parser in C):
#   X <name> <n> <ret> <lag> <mode> <bound>   then 2n source values, n sample values
#   P <name> <n> <mode> <value>               then n + n values
# mode: E coef == bound, G coef > bound, L coef < bound, N NaN / no check
with open(os.path.join(HERE, "reference_kat.txt"), "w") as f:
    for c in xc:
        e = c["expect"]
        mode, bound = "N", 0.0
        if "coef_eq" in e: mode, bound = "E", e["coef_eq"]
        if "coef_gt" in e: mode, bound = "G", e["coef_gt"]
        if "coef_lt" in e: mode, bound = "L", e["coef_lt"]
        f.write("X %s %d %d %d %s %r\n" % (c["name"], len(c["sample"]), e["ret"], e.get("lag", 0), mode, bound))
        f.write(" ".join(repr(float(v)) for v in c["source"]) + "\n")
        f.write(" ".join(repr(float(v)) for v in c["sample"]) + "\n")
    for c in pc:
        e = c["expect"]
        mode, bound = ("N", 0.0) if e.get("nan") else ("E", e["eq"])
        f.write("P %s %d %s %r\n" % (c["name"], len(c["source_seg"]), mode, bound))
        f.write(" ".join(repr(float(v)) for v in c["source_seg"]) + "\n")
        f.write(" ".join(repr(float(v)) for v in c["sample_seg"]) + "\n")
print("wrote", len(xc), "+", len(pc), "cases")
