import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model():
    from wb_humanoid_mpc_amd import load_model
    return load_model()


@pytest.fixture(scope="session")
def oracle(model):
    from hsqp_oracle import Oracle
    return Oracle(model)


@pytest.fixture(scope="session")
def cmodel():
    from wb_humanoid_mpc_amd import load_model
    return load_model(formulation="centroidal")


@pytest.fixture(scope="session")
def coracle(cmodel):
    from hsqp_oracle import Oracle
    return Oracle(cmodel)


@pytest.fixture()
def rng():
    return np.random.default_rng(1234)


def random_state_input(model, rng, scale=1.0):
    """A generic, well-conditioned (x,u): joints inside limits, moderate velocities/accelerations/wrenches."""
    nj = model.nj
    x = model.initial_state.copy()
    x[:3] += 0.05 * scale * rng.standard_normal(3)
    x[3:6] += 0.2 * scale * rng.standard_normal(3)
    x[6:6 + nj] = np.clip(x[6:6 + nj] + 0.3 * scale * rng.standard_normal(nj), model.q_lo + 0.1, model.q_hi - 0.1)
    x[6 + nj:] = 0.5 * scale * rng.standard_normal(6 + nj)
    u = np.zeros(model.nu)
    u[2] = u[8] = model.total_mass * 9.81 / 2
    u[:12] += 10.0 * scale * rng.standard_normal(12)
    u[12:] = 2.0 * scale * rng.standard_normal(nj)
    return x, u
