import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Tests marked `gpu` are SKIPPED (with the reason shown) on a machine without a HIP device instead of
    failing one by one inside svo_create; on a GPU box nothing is skipped."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible: the hot path has no CPU fallback (run on an MI355X)")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    O = entry.load_oracle()
    O.build()
    return O


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    import importlib
    return importlib.import_module(entry.PKG_NAME + ".synth")


@pytest.fixture(scope="session")
def small_seq(synth):
    """4 frames of a 416x128 corridor, rendered once per session (CPU, torch)."""
    seq = synth.StereoSequence(width=416, height=128, n_frames=4, seed=11)
    frames = [tuple(x.numpy() for x in seq.render(t)) for t in range(4)]
    return seq, frames


def rand_image(h, w, seed, blocks=True):
    """Random blocky test image with plenty of FAST corners."""
    rng = np.random.default_rng(seed)
    if blocks:
        bh, bw = (h + 5) // 6, (w + 5) // 6
        img = rng.integers(0, 256, (bh, bw), dtype=np.uint8).repeat(6, 0).repeat(6, 1)[:h, :w]
        noise = rng.integers(-6, 7, (h, w))
        return np.clip(img.astype(int) + noise, 0, 255).astype(np.uint8)
    return rng.integers(0, 256, (h, w), dtype=np.uint8)
