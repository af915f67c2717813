import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "movie-recommender-system_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (its directory name is not a Python identifier)."""
    return importlib.import_module(PKG_NAME)


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module(PKG_NAME + ".synth")


@pytest.fixture(scope="session")
def oracle():
    from oracle import knncf_oracle

    knncf_oracle.build()
    return knncf_oracle


@pytest.fixture(scope="session")
def syn100k(synth):
    return synth.syn_100k()
