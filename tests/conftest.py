import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MODELS = os.path.join(ROOT, "models")
GOLDEN = os.path.join(ROOT, "tests", "golden")

MODEL_FILES = {
    "back": "face_detection_back.tflite",
    "front": "face_detection_front.tflite",
    "short": "face_detection_short_range.tflite",
    "full": "face_detection_full_range.tflite",
    "sparse": "face_detection_full_range_sparse.tflite",
    "landmark": "face_landmark.tflite",
    "iris": "iris_landmark.tflite",
}
INPUT_RANGE = {"back": (-1, 1), "front": (-1, 1), "short": (-1, 1), "full": (-1, 1), "sparse": (-1, 1), "landmark": (0, 1), "iris": (0, 1)}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def model_path(name):
    return os.path.join(MODELS, MODEL_FILES[name])


def seeded_input(name, batch, seed, shape_hw):
    """Deterministic synthetic frames in the model's input range (numpy RandomState is stream-stable)."""
    lo, hi = INPUT_RANGE[name]
    rs = np.random.RandomState(seed)
    return rs.uniform(lo, hi, (batch, shape_hw[0], shape_hw[1], 3)).astype(np.float32)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def mi():
    import rs_face_detection_tflite_amd as m
    return m


@pytest.fixture(scope="session")
def man_image():
    from PIL import Image
    return np.asarray(Image.open(os.path.join(GOLDEN, "man.jpg")).convert("RGB"))
