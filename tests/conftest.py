import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden_params():
    with open(os.path.join(GOLDEN, 'params.json')) as fp:
        return json.load(fp)


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope='session')
def golden_kernels():
    return _load('kernels.npz')


@pytest.fixture(scope='session')
def golden_vmc_tape():
    return _load('vmc_tape.npz')


@pytest.fixture(scope='session')
def golden_dmc_tape():
    return _load('dmc_tape.npz')


@pytest.fixture(scope='session')
def golden_reblock():
    return _load('reblock.npz')


@pytest.fixture(scope='session')
def golden_stats():
    return _load('stats.npz')


@pytest.fixture(scope='session')
def oracle():
    """The CPU oracle (test infrastructure), built on demand."""
    from oracle import qmc_oracle
    qmc_oracle.build()
    return qmc_oracle


def oracle_model(oracle, golden_params, tag):
    rec = golden_params[tag]
    return oracle.model_from_params(rec['params'], rec['obf_params'],
                                    rec['tbf_params'])
