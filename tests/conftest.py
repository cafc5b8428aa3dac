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
    config.addinivalue_line("markers", "gate: the <= 3-minute GPU subset that validates a changed binary (pytest -m 'gpu and gate'; README)")
    # (r06) the deterministic weight filler is a pure function of (name, shape, profile): memoise it for the session -- the GPU suite
    # builds the same full-size models dozens of times at ~20 s of single-core numpy each
    from frido_amd import synth
    synth.enable_cache()

# The GPU suite runs on the tiles the BENCHMARK runs on: the committed, library-hash-keyed tile cache (profiles/tune_cache.json,
# written by `bench.py --retune`) is read-only here, so a GEMM signature the benchmark uses gets the benchmark's tile in every
# test process; signatures the cache does not hold take the library's static tile (FRIDO_TUNE_ON_MISS): nothing in the suite depends on
# what a live tuner happened to measure, so a flipped VQ code cannot come and go between runs.
os.environ.setdefault("FRIDO_TUNE_CACHE", os.path.join(REPO, "profiles", "tune_cache.json"))
os.environ.setdefault("FRIDO_TUNE_CACHE_READONLY", "1")
os.environ.setdefault("FRIDO_TUNE_ON_MISS", "static")       # no live (timing-dependent) tile choice inside the suite: its results are repeatable bit for bit


def pytest_sessionstart(session):
    """(r05, advisor) The pinned tile cache is keyed by a content hash of libfrido_hip.so; on a box whose build is not byte-identical
    EVERY lookup misses and -- with FRIDO_TUNE_ON_MISS=static -- the model-level tests silently run on the static tiles instead of
    the benchmark's.  Say so, loudly, once."""
    import json
    import warnings
    path = os.environ.get("FRIDO_TUNE_CACHE", "")
    if not path or not os.path.exists(path) or not os.path.exists(os.path.join(REPO, "frido_amd", "libfrido_hip.so")):
        return
    try:
        from frido_amd import tune
        tag, have = json.load(open(path)).get("lib"), tune._lib_tag()
    except Exception as e:      # noqa: BLE001
        warnings.warn(f"tests/conftest.py: cannot compare the tile cache's library tag ({e})")
        return
    if tag != have:
        msg = (f"tile cache {path} was pinned for libfrido_hip.so {tag}, this build is {have}: every GEMM of the GPU suite runs on the "
               "library's STATIC tile (no split-K, no deferred reductions) -- rebuild bit-identically or re-pin with `python bench.py --retune`")
        warnings.warn(msg)
        print("\nWARNING: " + msg, file=sys.stderr)
