"""CPU: the oracle against its committed golden vectors (tests/golden/make_golden.py).

Parity with the reference is unpinned (it has no vectors for this path and OpenCV is absent);
these vectors pin the oracle against itself and against the generators."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402


def test_oracle_reproduces_golden_vectors():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_golden.npz")
    ref = np.load(path, allow_pickle=False)
    got = make_golden.build()
    assert set(got) == set(ref.files)
    for k in ref.files:
        assert np.array_equal(np.asarray(got[k]), ref[k]), k
