# -*- coding: utf-8 -*-
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "pytorch-vae_amd")
GOLD = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG, GOLD):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLD, name + ".npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden
