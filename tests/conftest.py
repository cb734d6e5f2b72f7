import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fetal-mri-segmentation_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Hot path first: under `pytest -x` a failure in a late, stochastic row of SURVEY 8(f) must not hide the kernels the bench
# times.  Files not listed keep their alphabetical place behind the listed ones; CPU tests are not reordered among themselves.
GPU_FILE_ORDER = ["test_gpu_ops", "test_gpu_losses", "test_gpu_engine", "test_gpu_fullsize_parity", "test_gpu_val_dice", "test_gpu_model", "test_gpu_fullsize", "test_gpu_dist",
                  "test_gpu_integration_snippet", "test_gpu_augment", "test_gpu_postprocess", "test_gpu_tta", "test_gpu_adversarial"]


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(GPU_FILE_ORDER)}

    def key(item):
        stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(stem, len(rank) if stem.startswith("test_gpu") else -1)
    items.sort(key=key)          # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
