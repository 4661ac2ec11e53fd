import os, sys, json
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long-running (true-shape) checks')


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def load_golden_weights(tag):
    import torch
    z = load_npz(f'cfg{tag}_weights.npz')
    cfg = json.loads(bytes(z.pop('__config__')).decode())
    return cfg, {k: torch.from_numpy(v) for k, v in z.items()}


@pytest.fixture(scope='session', params=['A', 'B'])
def golden_model(request):
    """(tag, oracle config dict, weights dict, ops dict) for the tiny seeded reference models."""
    import torch
    cfg, w = load_golden_weights(request.param)
    ops = {k: torch.from_numpy(v) for k, v in load_npz(f'cfg{request.param}_ops.npz').items()}
    return request.param, cfg, w, ops
