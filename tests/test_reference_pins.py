"""Known answers produced by the IMPORTED reference (tests/golden/make_utils_golden.py -> tests/golden/utils_kat.json): W/utils.py's
Dequantize over all 256 byte values and GetListOfFeatureNamesAndSizes, W/inference.py's format_lines.  Together with the metric
KATs (tests/test_oracle_metrics.py) these are the hot-path functions the reference itself can vouch for in this container; every
other oracle function is pinned only against independent third-party implementations."""
import json
import logging
import os

import numpy as np
import pytest
import torch

from oracle import np_ref

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "utils_kat.json")))


def _table(key):
    return np.frombuffer(bytes.fromhex("".join(KAT[key]["hex"])), dtype=np.float32)


def test_fixture_is_what_the_reference_docstring_says():
    t = _table("dequantize_default")
    assert t.shape == (256,) and t[0] == np.float32(-1.9921875) and t[255] == np.float32(2.0078125)
    assert np.all(np.diff(t) > 0)
    assert KAT["dequantize_bad_range"] == "AssertionError"


def test_oracle_dequantize_equals_the_reference_table():
    q = np.arange(256)
    assert np.array_equal(np_ref.dequantize(q, dtype=np.float32), _table("dequantize_default"))            # bit for bit in fp32
    assert np.array_equal(np_ref.dequantize(q, 4.0, -1.0, dtype=np.float32), _table("dequantize_max4_min-1"))
    assert np.abs(np_ref.dequantize(q) - np.asarray(KAT["dequantize_default_f64"])).max() == 0.0
    with pytest.raises(AssertionError):
        np_ref.dequantize(q, 1.0, 1.0)
    # the folded form the fused kernels implement is the same affine map before the normalisation
    x = np_ref.dequant_l2norm_folded(q.astype(np.uint8)[None, None, :])
    ref = np.asarray(KAT["dequantize_default_f64"])
    assert np.abs(x[0, 0] - ref / np.sqrt((ref ** 2).sum())).max() < 1e-15


def test_product_feature_names_and_sizes_follow_the_reference(caplog):
    import yt8m_amd.utils as utils
    for c in KAT["feature_names_and_sizes"]:
        if "raises" in c:
            with pytest.raises(ValueError):
                utils.GetListOfFeatureNamesAndSizes(c["names"], c["sizes"])
            continue
        caplog.clear()
        with caplog.at_level(logging.ERROR):
            names, sizes = utils.GetListOfFeatureNamesAndSizes(c["names"], c["sizes"])
        assert [names, sizes] == c["result"]
        assert len([r for r in caplog.records if r.levelno >= logging.ERROR]) == c["logged_errors"]


def _format_case(c):
    rs = np.random.RandomState(c["seed"])
    p = np.stack([(rs.permutation(c["V"]) + rs.rand()) / c["V"] for _ in range(c["B"])]).astype(np.float32)
    ids = [("vid%04d" % ((c["seed"] - 500) * 10 + i)).encode("utf-8") for i in range(c["B"])]
    return ids, p


def test_product_format_lines_host_path_equals_the_reference_lines():
    import yt8m_amd.inference as inference
    for c in KAT["format_lines"]:
        ids, p = _format_case(c)
        assert list(inference.format_lines(ids, p, c["top_k"])) == c["lines"]


@pytest.mark.gpu
def test_device_dequantize_equals_the_reference_table(dev):
    import yt8m_amd.ops as ops
    import yt8m_amd.utils as utils
    q = torch.arange(256, dtype=torch.uint8, device=dev)
    got = utils.Dequantize(q).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, _table("dequantize_default"))                    # bit for bit
    got = utils.Dequantize(q, 4, -1).cpu().numpy()
    assert np.array_equal(got, _table("dequantize_max4_min-1"))
    # the fused reader-side kernel (yt8m_dequant_l2norm_u8: dequantise + zero padding + l2-normalise, W/readers.py:178-187 +
    # default_transformer.py:4-8): one frame holding every byte value, one frame beyond num_frames
    ref = np.asarray(KAT["dequantize_default_f64"])
    frames = torch.stack([q, q.flip(0), q]).view(1, 3, 256).contiguous()
    nf = torch.tensor([2], dtype=torch.int32, device=dev)
    x = ops.dequant_l2norm(frames, nf).cpu().numpy().astype(np.float64)
    unit = ref / np.sqrt((ref ** 2).sum())
    assert np.abs(x[0, 0] - unit).max() < 2e-7 and np.abs(x[0, 1] - unit[::-1]).max() < 2e-7
    assert np.all(x[0, 2] == 0.0)
    # video-level reader contract ("average of dequantized values", W/readers.py:69-71) through the mean kernel
    m = ops.dequant_mean_l2norm(frames, nf).cpu().numpy().astype(np.float64)
    mean = (ref + ref[::-1]) / 2
    assert np.abs(m[0] - mean / np.sqrt((mean ** 2).sum())).max() < 2e-7


@pytest.mark.gpu
def test_device_format_lines_equals_the_reference_lines(dev):
    import yt8m_amd.inference as inference
    for c in KAT["format_lines"]:
        ids, p = _format_case(c)
        if c["top_k"] > c["V"]:
            continue
        assert list(inference.format_lines(ids, torch.from_numpy(p).to(dev), c["top_k"])) == c["lines"], c["seed"]
