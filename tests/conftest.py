import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TRACKS = {"big_track": os.path.join(ROOT, "tracks", "big_track.json"),
          "track": os.path.join(ROOT, "tracks", "track.json"),
          "oval64": os.path.join(ROOT, "tracks", "oval64.json")}    # generated (track_tool make-oval): 128 walls, 40 gates
ENV_CONFIGS = [(t, n) for t in ("big_track", "track") for n in (12, 16, 32)] + [("oval64", 16)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
