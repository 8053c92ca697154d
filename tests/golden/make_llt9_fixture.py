#!/usr/bin/env python3
"""Writes tests/golden/llt9_ill_conditioned.json: the only numeric known-answer the reference
holds for this path -- the inputs (lower-triangular float32 JtJ, Jtr, lambda) and the two printed
outputs of cpp/examples/levmarq_ill_conditioned_float32_issue.cpp:16-63.  The numbers below are
DATA transcribed from that example (values found by its author while tracking a geometry)."""
import json
import os

JTJ_LOWER = [
    [557551.4375],
    [296441.21875, 657639.8125],
    [-4293.4072265625, -5085.32958984375, 364752.1875],
    [42399.52734375, 131392.296875, 31440.83984375, 70597.6328125],
    [-27725.1328125, 44876.76953125, -105931.8828125, 0.0, 70597.6328125],
    [-43429.875, -83350.875, -62037.90625, -55166.17578125, 25584.125, 52518.796875],
    [1993.02294921875, 3831.88916015625, 2867.069091796875, 2574.660400390625, -1193.505981445312, -2450.312255859375,
     114.358093261719],
    [1947.6396484375, 6048.806640625, 1454.197631835938, 3295.457763671875, 0.0, -2574.660400390625, 120.201538085938,
     153.880432128906],
    [-1262.969848632812, 2073.468505859375, -4891.965820312500, 0.0, 3295.457763671875, 1193.505981445312,
     -55.693786621094, 0.0, 153.880432128906],
]
JTR = [-2.338238716125, -4.207848548889, 3.598472595215, -1.105026721954, -1.491069078445, 0.368796110153,
       -0.017316624522, -0.051174595952, -0.068564474583]

out = {
    "source": "cpp/examples/levmarq_ill_conditioned_float32_issue.cpp:16-63",
    "lambda": 1.5607382e-06,
    "jtj_lower_rows": JTJ_LOWER,
    "jtr": JTR,
    "reference_float32_residual_norm": 0.0028946274,
    "reference_float32_expected_cost_change": 0.000244110823,
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "llt9_ill_conditioned.json"), "w") as f:
    json.dump(out, f, indent=1)
