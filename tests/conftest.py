import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def experiments_loaded():
    """True when this process runs on libaae_hip_experiments.so (AAE_EXPERIMENTS=1 in the environment and the library built:
    `python __graft_entry__.py experiments`).  The product library carries only the kernel forms the planner uses; tests that
    compare those with the variants that measured slower (8-wave / 3-slab wave-split-K, the persistent per-detection launch,
    register-staged igemm, the round-1 scans ...) need the experiments build and are skipped without it."""
    from augmentedautoencoder_amd import _lib
    return _lib.experiments_requested() and os.path.exists(_lib.library_path())


needs_experiments = pytest.mark.skipif(not experiments_loaded(), reason='needs the experiments build: AAE_EXPERIMENTS=1 + `python __graft_entry__.py experiments`')


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """GPU parity runs leave their measured flip counts / gaps / layer errors in gpurun_out/parity_report.json."""
    try:
        import parity_report
        parity_report.dump()
    except Exception:
        pass
