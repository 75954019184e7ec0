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


@pytest.fixture(autouse=True)
def honour_lstm_chunks():
    """Tests parametrise the time partition of the LSTM stack (multi-launch recurrences, t0 > 0 paths): make the persistent path
    follow the caller's `chunks` as the per-step path does.  The product's own partition (1 forward chunk, 3 backward parts) is
    exercised by test_gpu_x3.py::test_persistent_partition_defaults and by bench.py / smoke()."""
    import yt8m_amd.seq_ops as seq_ops
    saved = seq_ops.PERSIST_FWD_CHUNKS, seq_ops.PERSIST_BWD_CHUNKS
    seq_ops.PERSIST_FWD_CHUNKS = seq_ops.PERSIST_BWD_CHUNKS = 0
    yield
    seq_ops.PERSIST_FWD_CHUNKS, seq_ops.PERSIST_BWD_CHUNKS = saved
