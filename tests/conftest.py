import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def _host_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return min(n, 32)


def pytest_configure(config):
    import torch
    torch.set_num_threads(_host_cores())     # the GPU box shows 256 CPUs but grants a 16-core share
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    path = os.path.join(REPO, "tests", "golden", "genconvit_b4.npz")
    return dict(np.load(path, allow_pickle=False))


@pytest.fixture(scope="session")
def hf_golden():
    """Outputs of the independent Hugging Face ConvNeXt / Swin implementations (tests/golden/make_hf_golden.py)."""
    import numpy as np
    return dict(np.load(os.path.join(REPO, "tests", "golden", "hf_backbones.npz"), allow_pickle=False))


_SD_CACHE = {}


def synthetic_sd(which: str):
    """Session-cached synthetic state dicts (the VAE one is 2.6 GB: build once)."""
    from genconvit_amd import spec, synth
    if which not in _SD_CACHE:
        if which == "ed":
            _SD_CACHE[which] = synth.make_state_dict(spec.ed_spec(), synth.DEFAULT_SEED, "ed/")
        elif which == "vae":
            _SD_CACHE[which] = synth.make_state_dict(spec.vae_spec(), synth.DEFAULT_SEED, "vae/")
        else:
            raise KeyError(which)
    return _SD_CACHE[which]


@pytest.fixture(scope="session")
def sd_ed():
    return synthetic_sd("ed")


@pytest.fixture(scope="session")
def sd_vae():
    return synthetic_sd("vae")
