import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU check")

# The GPU suite runs on the tiles the BENCHMARK runs on: the committed, library-hash-keyed tile cache (profiles/tune_cache.json,
# written by `bench.py --retune`) is read-only here, so a GEMM signature the benchmark uses gets the benchmark's tile in every
# test process; signatures the cache does not hold take the library's static tile (FRIDO_TUNE_ON_MISS): nothing in the suite depends on
# what a live tuner happened to measure, so a flipped VQ code cannot come and go between runs.
os.environ.setdefault("FRIDO_TUNE_CACHE", os.path.join(REPO, "profiles", "tune_cache.json"))
os.environ.setdefault("FRIDO_TUNE_CACHE_READONLY", "1")
os.environ.setdefault("FRIDO_TUNE_ON_MISS", "static")       # no live (timing-dependent) tile choice inside the suite: its results are repeatable bit for bit
