import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def toy():
    """(opts, taxo, hash, genomes, Taxonomy) of the toy multi-taxon database."""
    from tests import synth
    return synth.toy_db()


@pytest.fixture(scope="session")
def toy_oracle(toy):
    from oracle import oracle as orc
    ob, tb, hb, _, _ = toy
    return orc.OracleDB(ob, tb, hb)


@pytest.fixture(scope="session")
def toy_engine(toy):
    from nohuman_amd import Engine
    ob, tb, hb, _, _ = toy
    eng = Engine.from_images(ob, tb, hb)
    yield eng
    eng.close()
