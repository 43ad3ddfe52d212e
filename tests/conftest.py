import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    out = {k: d[k] for k in d.files if k != "meta"}
    out["meta"] = json.loads(str(d["meta"]))
    return out


@pytest.fixture(scope="session")
def xv_weights():
    from speakerguard_amd import synth
    return synth.make_xv_weights(seed=0, D=200, n_spk=10)


def weights_checksum(w):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(w["state_dict"]):
        h.update(np.ascontiguousarray(w["state_dict"][k]).tobytes())
    for k in ("emb_mean", "lda", "plda_mean", "plda_transform", "plda_psi", "enroll"):
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()


PARITY_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_log.txt")


def log(msg):
    """Measured parity numbers of the GPU tests (copied to profiles/ at the end of a round)."""
    os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
    with open(PARITY_LOG, "a") as f:
        f.write(msg + "\n")
    print(msg)
