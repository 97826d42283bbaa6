import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def micro_cfg():
    from whisper_mojo_amd import WhisperConfig
    return WhisperConfig.micro()


@pytest.fixture(scope="session")
def tiny_cfg():
    from whisper_mojo_amd import WhisperConfig
    return WhisperConfig.tiny()


@pytest.fixture(scope="session")
def micro_weights(micro_cfg):
    from whisper_mojo_amd import synth
    return synth.synth_weights(micro_cfg, 0)


@pytest.fixture(scope="session")
def tiny_weights(tiny_cfg):
    # the C generator is ~40x faster than numpy for the 151 MB image; test_synth.py proves they are equal
    from oracle import oracle
    return oracle.synth_weights_c(tiny_cfg, 0)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))
