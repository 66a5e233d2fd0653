"""Random layer shapes (both kernel families): forward, masked dgrad and wgrad through the C ABI against the oracle."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _fuzz():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'scripts', 'fuzz_conv.py')
    spec = importlib.util.spec_from_file_location('fuzz_conv', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize('path', [1, 0], ids=['pipelined', 'two_wg_per_cu'])
def test_random_shapes_vs_oracle(path):
    from ml_super_resolution_amd import _lib
    old = _lib.lib().srx_set_conv_path(path)
    try:
        failures = _fuzz().run(60, 1000 + path, verbose=False)
    finally:
        _lib.lib().srx_set_conv_path(old)
    assert not failures, failures[:5]
