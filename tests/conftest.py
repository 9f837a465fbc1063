"""pytest configuration: markers + shared fixture loaders."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
	return torch.load(os.path.join(GOLDEN, name), weights_only=False)


@pytest.fixture(scope="session")
def golden():
	return load_golden


def has_gpu():
	return torch.cuda.is_available()


@pytest.fixture(autouse=True)
def _pin_global_rng(request):
	"""Every test starts from the same global torch RNG state (CPU and, when present, GPU): inputs drawn without an explicit generator are the
	same on every run, so a pass here is a pass at review time."""
	import zlib
	seed = 0x5EED ^ (zlib.crc32(request.node.nodeid.encode()) & 0xFFFF)  # str hashes are salted per process; crc32 is not
	torch.manual_seed(seed)
	yield
