"""The randomised device-vs-oracle sweeps of tools/ (fixed seeds, sized for <= 30 s each) so that the driver's
`pytest -m gpu` runs them too."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("seed", [2026, 7])
def test_vardct_fuzz(seed):
    assert _load("fuzz_gpu").run(n_cases=12, seed=seed, verbose=False) == 0


@pytest.mark.parametrize("seed", [11, 12])
def test_modular_fuzz(seed):
    assert _load("fuzz_modular_gpu").run(n_cases=40, seed=seed, verbose=False) == 0
