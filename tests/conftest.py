import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__  # noqa: E402

__graft_entry__.load_package()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture()
def flags():
    from yt8m_amd.flags import FLAGS
    FLAGS.reset()
    yield FLAGS
    FLAGS.reset()


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.set_device(0)
    return torch.device("cuda:0")


def _set_partition(mode):
    import yt8m_amd.seq_ops as seq_ops
    saved = seq_ops.PERSIST_FWD_CHUNKS, seq_ops.PERSIST_BWD_CHUNKS
    if mode == "caller":
        seq_ops.PERSIST_FWD_CHUNKS = seq_ops.PERSIST_BWD_CHUNKS = 0
    return saved


@pytest.fixture()
def honour_lstm_chunks():
    """For tests whose SUBJECT is the time partition of the LSTM stack (multi-launch recurrences, t0 > 0 paths, launch counts):
    makes the persistent path follow the caller's `chunks` as the per-step path does.  Not autouse: every other test
    runs the product's own partition (1 forward launch per layer, 3 backward parts -- what bench.py and smoke() run)."""
    import yt8m_amd.seq_ops as seq_ops
    saved = _set_partition("caller")
    yield
    seq_ops.PERSIST_FWD_CHUNKS, seq_ops.PERSIST_BWD_CHUNKS = saved


@pytest.fixture(params=["product-partition", "callers-chunks"])
def lstm_partition(request):
    """Model-level LSTM tests run twice: on the product's default time partition and on the caller's `chunks`."""
    import yt8m_amd.seq_ops as seq_ops
    saved = _set_partition("caller" if request.param == "callers-chunks" else "product")
    yield request.param
    seq_ops.PERSIST_FWD_CHUNKS, seq_ops.PERSIST_BWD_CHUNKS = saved
