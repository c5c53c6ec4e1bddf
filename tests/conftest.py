import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_unconfigure(config):
    """Composable-kernel instances inside MIOpen / hipBLASLt print a "GridwiseOp: Problemsize ..." line per call
    into the C stdio buffer, which is flushed when the process exits -- AFTER pytest's summary, so that the tail
    of a run's output (what the driver records) is nothing but that chatter.  Only when a GPU was used (on CPU runs
    nothing prints there), and as late as possible -- an atexit hook registered here, i.e. behind pytest's own
    reporting, plugins' unconfigure hooks and in-process callers that print after pytest.main() -- the buffer is
    flushed into /dev/null; the temporary descriptor is closed again."""
    if not (torch.cuda.is_available() and torch.cuda.is_initialized()):
        return
    import atexit
    import ctypes

    def _drop_c_stdio():
        try:
            sys.stdout.flush()
            fd = os.open(os.devnull, os.O_WRONLY)
            os.dup2(fd, 1)
            os.close(fd)
            ctypes.CDLL(None).fflush(None)
        except Exception:           # noqa: BLE001 -- never fail a run over its log
            pass

    atexit.register(_drop_c_stdio)


def load_golden(name):
    """Load a committed fixture as a dict of torch tensors."""
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def tiny_common():
    return load_golden("tiny_common.npz")


RENDER_VARIANTS = [("sdf", False), ("sdf", True), ("naive", False), ("naive", True)]


def render_fixture_name(mode, cat_seg):
    return f"tiny_render_{mode}_{'catseg' if cat_seg else 'plain'}.npz"
