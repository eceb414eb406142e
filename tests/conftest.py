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
    # a fresh checkout has no built library (it is git-ignored): build it once, like the driver's build() step.
    # hipcc cross-compiles gfx950 without a GPU; if the toolchain is missing the tests that need the library fail loudly.
    lib = os.path.join(PKG, "vqvae_hip", "libvqvae_hip.so")
    if not os.path.isfile(lib):
        import shutil
        if shutil.which("hipcc") or os.path.isfile("/opt/rocm/bin/hipcc"):
            import __graft_entry__ as entry
            entry.build()


def load_golden(name):
    path = os.path.join(GOLD, name + ".npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden
