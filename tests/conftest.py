import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


class Fixture:
    """One tests/golden/*.npz: `.meta` (dict) + arrays by name (torch tensors on CPU)."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(z["meta_json"]).decode())
        self._z = z

    def __contains__(self, k):
        return k in self._z.files

    def np(self, k):
        return self._z[k]

    def t(self, k):
        a = self._z[k]
        return torch.from_numpy(np.ascontiguousarray(a)).reshape(a.shape)

    def keys(self):
        return [k for k in self._z.files if k != "meta_json"]

    def batch(self):
        b = {}
        for k in ("context_x", "context_y", "query_x", "query_y", "target_x", "target_y",
                  "target_theta", "target_all", "target_mask"):
            if "in_" + k in self:
                b[k] = self.t("in_" + k)
        return b

    def cfg(self):
        d = self.meta["dims"]
        return dict(embedding_type=d["embedding_type"], n_head=d["n_head"], num_layers=d["L"],
                    num_components=d["C"], std_min=1e-4, time_token=d.get("time_token", False),
                    n_target_theta=d["n_theta"])

    def forced_idx(self, mode="train"):
        T = self.meta["T"]
        return torch.cat([self.t(f"{mode}.idx_{t}") for t in range(T)], dim=1)


MODEL_FIXTURES = ["cfg2_location_d32", "cfg2_location_d256", "cfg1_almix_d1_data",
                  "cfg1_almix_d1_theta", "cfg1_almix_d1_all", "cfg3_almix_d2", "cfg4_ces",
                  "cfg5_psycho_d512", "aux_data_timetoken", "aux_timetoken_train"]


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Fixture(name)
        return cache[name]
    return get
