import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """-> (meta dict, arrays dict) of a fixture written by oracle/make_golden.py."""
    with np.load(os.path.join(GOLD, name + ".npz")) as z:
        arrays = {k: z[k] for k in z.files if k != "meta"}
        meta = json.loads(bytes(z["meta"]).decode())
    return meta, arrays


# Every GPU test module runs once per arithmetic mode of the library (include/cld.h CLD_PRECISION_*): the exact-fp32 MFMA
# default and the optional f16x2 split mode, same parity bars.  CLD_TEST_PRECISION=f32|f16x2 narrows a run to one mode.
_PRECISIONS = [os.environ["CLD_TEST_PRECISION"]] if os.environ.get("CLD_TEST_PRECISION") else ["f32", "f16x2"]


@pytest.fixture(scope="module", params=_PRECISIONS)
def precision(request):
    return request.param


@pytest.fixture(scope="session")
def golden():
    return load_golden
