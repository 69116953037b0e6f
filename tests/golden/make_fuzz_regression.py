"""Writes tests/golden/fuzz_regression_features_20_4.npz: the CLASS of input behind the round-2 development mismatch of
hk_get_features (tests/fuzz_parity.py operator_case, kind 1: m=20, d=4, b=63, dyadic fractions, scale=True; found and
fixed while qd_ranks_stable was being written, never committed as a case) -- seeded inputs of that class and the
oracle's features for them.  Test infrastructure: run from the repo root, `python tests/golden/make_fuzz_regression.py`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import c_oracle as CO  # noqa: E402

rng = np.random.default_rng(20040063)
b, m, d = 63, 20, 4
p = rng.integers(0, 6, (b, m, d)).astype(np.float32)
p[rng.random((b, m)) < 0.3] = -1.0
p = np.where(p >= 0, p / np.float32(4.0), p).astype(np.float32)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_regression_features_20_4.npz")
np.savez_compressed(out, points=p, features_scaled=CO.get_features(p, True), features_unscaled=CO.get_features(p, False))
print(out, p.shape)
