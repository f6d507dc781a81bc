import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """(meta dict, npz arrays) of one fixture written by tools/gen_golden.py."""
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return meta, {k: z[k] for k in z.files}


ENV_GOLDENS = ["env_bench_shape", "env_l1_ladder4", "env_bufferfull_i05", "env_starved_i03",
               "env_speed125", "env_const_policy", "env_l3_i07"]
MPC_GOLDENS = ["mpc_b6h5_cbr", "mpc_b6h5_vbr", "mpc_b4h5_l1", "mpc_b6h3_smallbuf", "mpc_b3h2",
               "mpc_b5h4", "mpc_b6h4_prevneg"]


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o
