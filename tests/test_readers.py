"""CPU tests of the native TFRecord / Example reader (csrc/tfrecord.hip via the C ABI) against the pure-Python
restatement and writer in oracle/tfrecord_ref.py.  Byte / integer work: every comparison is bit-exact."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import tfrecord_ref as tr
import yt8m_amd._lib as L
import yt8m_amd.readers as readers


def test_crc32c_known_answers():
    """RFC 3720 B.4 check values pin the oracle; the native CRC (slicing-by-8) must match it on ragged lengths."""
    assert tr.crc32c(b"123456789") == 0xE3069283
    assert tr.crc32c(bytes(32)) == 0x8A9136AA
    assert tr.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert tr.crc32c(bytes(range(32))) == 0x46DD794E
    lib = L.lib()
    rs = np.random.RandomState(0)
    for n in [0, 1, 7, 8, 9, 63, 64, 65, 1000, 4097]:
        b = rs.randint(0, 256, size=n).astype(np.uint8).tobytes()
        buf = ctypes.create_string_buffer(b, len(b))
        assert lib.yt8m_crc32c(buf, len(b)) == tr.crc32c(b)
        assert lib.yt8m_crc32c_masked(buf, len(b)) == tr.masked_crc(b)


def _videos(rs, n, names, sizes, max_len):
    vids = []
    for i in range(n):
        k = int(rs.randint(1, max_len + 1))
        vids.append(dict(video_id=("vid%04d" % i).encode(), labels=sorted(set(rs.randint(0, 4716, size=rs.randint(0, 6)).tolist())),
                         frames={nm: rs.randint(0, 256, size=(k, s)).astype(np.uint8) for nm, s in zip(names, sizes)},
                         features={("mean_" + nm): rs.randn(s).astype(np.float32) for nm, s in zip(names, sizes)}))
    return vids


@pytest.mark.parametrize("packed", [True, False])
def test_frame_reader_bit_exact(tmp_path, packed):
    rs = np.random.RandomState(1)
    names, sizes = ["rgb", "audio"], [1024, 128]
    vids = _videos(rs, 11, names, sizes, max_len=40)
    vids[3]["frames"] = {nm: rs.randint(0, 256, size=(45, s)).astype(np.uint8) for nm, s in zip(names, sizes)}   # > max_frames: truncated
    vids[5]["labels"] = [7, 7, 3, 4715]                                                                          # duplicates
    vids[6]["labels"] = []                                                                                       # no labels
    p1, p2 = str(tmp_path / "a.tfrecord"), str(tmp_path / "b.tfrecord")
    tr.write_frame_shard(p1, vids[:7], names, packed)
    tr.write_frame_shard(p2, vids[7:], names, packed)
    rd = readers.YT8MFrameFeatureReader(num_classes=4716, feature_sizes=sizes, feature_names=names, max_frames=30)
    got = list(rd.prepare_reader([p1, p2], batch_size=4))
    assert [len(b[0]) for b in got] == [4, 3, 4]                       # per-file read_up_to semantics
    ids = [i for b in got for i in b[0]]
    q = torch.cat([b[1] for b in got]).numpy()
    lab = torch.cat([b[2] for b in got]).numpy()
    nf = torch.cat([b[3] for b in got]).numpy()
    eq, enf, elab = tr.expected_frame_batch(vids, names, sizes, 30, 4716)
    assert ids == [v["video_id"] for v in vids]
    assert np.array_equal(q, eq) and np.array_equal(nf, enf) and np.array_equal(lab, elab)
    assert q.dtype == np.uint8 and nf[3] == 30 and lab[5].sum() == 3 and lab[6].sum() == 0
    # glob pattern form + IOError for no match (W/train.py:193-195)
    assert sum(len(b[0]) for b in rd.prepare_reader(str(tmp_path / "*.tfrecord"), batch_size=64)) == 11
    with pytest.raises(IOError):
        list(rd.prepare_reader(str(tmp_path / "nope*.tfrecord")))


def test_video_reader_bit_exact(tmp_path):
    rs = np.random.RandomState(2)
    names, sizes = ["mean_rgb", "mean_audio"], [1024, 128]
    vids = _videos(rs, 9, ["rgb", "audio"], sizes, max_len=3)
    p = str(tmp_path / "v.tfrecord")
    tr.write_video_shard(p, vids, names)
    rd = readers.YT8MAggregatedFeatureReader(num_classes=4716, feature_sizes=sizes, feature_names=names)
    got = list(rd.prepare_reader(p, batch_size=1024))
    assert len(got) == 1
    ids, x, lab, ones = got[0]
    ex = np.stack([np.concatenate([v["features"][n] for n in names]) for v in vids])
    assert np.array_equal(x.numpy(), ex) and x.dtype == torch.float32 and ids[8] == b"vid0008"
    for i, v in enumerate(vids):
        assert sorted(np.nonzero(lab[i].numpy())[0].tolist()) == sorted(set(v["labels"]))
    assert torch.equal(ones, torch.ones(9))


def test_reader_errors(tmp_path):
    rs = np.random.RandomState(3)
    names, sizes = ["rgb"], [16]
    vids = _videos(rs, 3, names, sizes, max_len=5)
    p = str(tmp_path / "f.tfrecord")
    tr.write_frame_shard(p, vids, names)
    raw = bytearray(open(p, "rb").read())
    raw[40] ^= 0xFF                                                  # flip a payload byte -> CRC mismatch
    bad = str(tmp_path / "bad.tfrecord")
    open(bad, "wb").write(bytes(raw))
    rd = readers.YT8MFrameFeatureReader(num_classes=4716, feature_sizes=sizes, feature_names=names, max_frames=8)
    with pytest.raises(ValueError, match="crc"):
        list(rd.prepare_reader(bad))
    assert len(list(rd.prepare_reader(p))[0][0]) == 3
    with pytest.raises(ValueError, match="missing"):                 # feature name not in the file
        list(readers.YT8MFrameFeatureReader(4716, [16], ["audio"], 8).prepare_reader(p))
    with pytest.raises(ValueError, match="bytes, expected"):         # wrong feature size
        list(readers.YT8MFrameFeatureReader(4716, [8], ["rgb"], 8).prepare_reader(p))
    open(str(tmp_path / "trunc.tfrecord"), "wb").write(open(p, "rb").read()[:-3])
    with pytest.raises(ValueError, match="truncated"):
        list(rd.prepare_reader(str(tmp_path / "trunc.tfrecord")))
    # the two feature lists of one video must have the same number of frames (tf.assert_equal, W/readers.py:239)
    two = [dict(video_id=b"x", labels=[1], frames={"rgb": rs.randint(0, 256, size=(4, 16)).astype(np.uint8),
                                                   "audio": rs.randint(0, 256, size=(5, 4)).astype(np.uint8)})]
    tr.write_frame_shard(str(tmp_path / "two.tfrecord"), two, ["rgb", "audio"])
    with pytest.raises(ValueError, match="disagree"):
        list(readers.YT8MFrameFeatureReader(4716, [16, 4], ["rgb", "audio"], 8).prepare_reader(str(tmp_path / "two.tfrecord")))
    with pytest.raises(AssertionError):
        readers.YT8MFrameFeatureReader(4716, [16, 4], ["rgb"], 8)
    with pytest.raises(NotImplementedError):
        readers.BaseReader().prepare_reader(None)


def test_prediction_dump_bit_exact_and_round_trip(tmp_path):
    """the ensemble-stage prediction dump (W/inference-pre-ensemble.py:291-308).  The native writer's
    file equals the oracle's pure-Python encoding byte for byte, and the native reader reads it back exactly."""
    import yt8m_amd.inference as inference
    rs = np.random.RandomState(5)
    n, V = 7, 4716
    pred = rs.rand(n, V).astype(np.float32)
    lab = rs.rand(n, V) < 0.001
    lab[3] = False                                                             # a video without labels
    ids = [("vid%03d" % i).encode() for i in range(n)]
    ids[2] = b"x"
    p = str(tmp_path / "predictions-0000.tfrecord")
    inference.write_to_record(p, ids, lab, pred)
    videos = [dict(video_id=ids[i], labels=list(np.nonzero(lab[i])[0]), features={"predictions": pred[i]}) for i in range(n)]
    q = str(tmp_path / "oracle.tfrecord")
    tr.write_video_shard(q, videos, ["predictions"])
    assert open(p, "rb").read() == open(q, "rb").read()
    rd = readers.YT8MAggregatedFeatureReader(num_classes=V, feature_sizes=[V], feature_names=["predictions"])
    got = list(rd.prepare_reader(p, batch_size=4))
    vids = [v for b in got for v in b[0]]
    x = np.concatenate([b[1].numpy() for b in got])
    y = np.concatenate([b[2].numpy() for b in got])
    assert vids == ids and np.array_equal(x, pred) and np.array_equal(y != 0, lab)
    with pytest.raises(ValueError):
        inference.write_to_record(str(tmp_path / "no_such_dir" / "f.tfrecord"), ids, lab, pred)


@pytest.mark.parametrize("threads", [1, 3])
def test_prefetcher_delivers_every_record_once(tmp_path, threads):
    """Native multi-threaded shard prefetcher (W/train.py:199-209 reader threads): with one thread the batch sequence equals
    the sequential reader's; with several, every record of every shard arrives exactly once, bit-exact; decode errors in a
    worker surface to the consumer."""
    rs = np.random.RandomState(9)
    names, sizes = ["rgb", "audio"], [16, 4]
    shards, allv = [], []
    for s, nrec in enumerate([5, 8, 1, 11, 4]):
        vids = _videos(rs, nrec, names, sizes, max_len=9)
        for i, v in enumerate(vids):
            v["video_id"] = ("s%dv%02d" % (s, i)).encode()
        p = str(tmp_path / ("f%d.tfrecord" % s))
        tr.write_frame_shard(p, vids, names)
        shards.append(p)
        allv += vids
    empty = str(tmp_path / "empty.tfrecord")                                   # a shard with no records is skipped cleanly
    open(empty, "wb").close()
    shards.insert(2, empty)
    rd = readers.YT8MFrameFeatureReader(num_classes=50, feature_sizes=sizes, feature_names=names, max_frames=6)
    seq = list(rd.prepare_reader(shards, batch_size=4))
    got = list(rd.prepare_reader(shards, batch_size=4, num_threads=threads, queue_depth=2))
    def flat(batches):
        out = {}
        for ids, q, lab, nf in batches:
            for i, v in enumerate(ids):
                assert v not in out
                out[v] = (q[i].numpy().tobytes(), lab[i].numpy().tobytes(), int(nf[i]))
        return out
    a, b = flat(seq), flat(got)
    assert len(a) == len(allv) and a == b
    if threads == 1:
        assert [ids for ids, _, _, _ in seq] == [ids for ids, _, _, _ in got]
    # video-level form
    vv = [dict(video_id=("x%d" % i).encode(), labels=[i % 7], features={"mean_rgb": rs.rand(8).astype(np.float32)}) for i in range(10)]
    p = str(tmp_path / "v.tfrecord")
    tr.write_video_shard(p, vv, ["mean_rgb"])
    rv = readers.YT8MAggregatedFeatureReader(num_classes=9, feature_sizes=[8], feature_names=["mean_rgb"])
    gv = list(rv.prepare_reader([p], batch_size=4, num_threads=threads))
    assert [len(b[0]) for b in gv] == [4, 4, 2]
    assert np.array_equal(np.concatenate([b[1].numpy() for b in gv]), np.stack([v["features"]["mean_rgb"] for v in vv]))
    # a corrupt shard: the worker's error reaches the consumer as an exception
    bad = str(tmp_path / "bad.tfrecord")
    raw = bytearray(open(shards[1], "rb").read())
    raw[40] ^= 0xFF
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        list(rd.prepare_reader([shards[0], bad], batch_size=4, num_threads=threads))


# ---- records encoded by the protobuf runtime (tests/golden/make_tfrecord_golden.py), not by this repository's writer --------
def _golden(golden_dir):
    import json
    import os
    meta = json.load(open(os.path.join(golden_dir, "tfrecord_golden.json")))
    return meta, os.path.join(golden_dir, "tfrecord_frame.bin"), os.path.join(golden_dir, "tfrecord_video.bin")


def test_protobuf_encoded_frame_records(golden_dir):
    """tf.train.SequenceExample bytes from google.protobuf (deterministic serialisation, maps sorted by key -- a different field
    order than the oracle writer emits) through the native reader: ids, labels (bit-exact multi-hot), num_frames, truncation to
    max_frames = 10 and the uint8 frame bytes (checksums + edge rows) all as the generator recorded them."""
    meta, frame_bin, _ = _golden(golden_dir)
    rd = readers.YT8MFrameFeatureReader(num_classes=meta["num_classes"], feature_sizes=meta["sizes"], feature_names=meta["names"],
                                        max_frames=meta["max_frames"])
    got = list(rd.prepare_reader([frame_bin], batch_size=16))
    assert len(got) == 1
    ids, q, lab, nf = got[0]
    q, lab, nf = q.numpy(), lab.numpy(), nf.numpy()
    assert q.dtype == np.uint8 and q.shape == (4, 10, 1152)
    for i, v in enumerate(meta["videos"]):
        assert ids[i] == v["video_id"].encode()
        assert sorted(np.nonzero(lab[i])[0].tolist()) == sorted(set(v["labels"]))
        k = min(v["num_frames"], 10)
        assert nf[i] == k and not q[i, k:].any()
        off = 0
        for n, s in zip(meta["names"], meta["sizes"]):
            blk = q[i, :k, off:off + s]
            assert blk[0, :8].tolist() == v["frame_first_row"][n] and blk[k - 1, -8:].tolist() == v["frame_last_row"][n]
            if v["num_frames"] <= 10:
                assert int(blk.astype(np.uint64).sum()) == v["frame_checksum"][n]
            off += s


def test_protobuf_encoded_video_records(golden_dir):
    meta, _, video_bin = _golden(golden_dir)
    names = ["mean_" + n for n in meta["names"]]
    rd = readers.YT8MAggregatedFeatureReader(num_classes=meta["num_classes"], feature_sizes=meta["sizes"], feature_names=names)
    got = list(rd.prepare_reader(video_bin, batch_size=16))
    ids, x, lab, _ = got[0]
    x, lab = x.numpy(), lab.numpy()
    for i, v in enumerate(meta["videos"]):
        assert ids[i] == v["video_id"].encode()
        assert sorted(np.nonzero(lab[i])[0].tolist()) == sorted(set(v["labels"]))
        off = 0
        for n, s in zip(names, meta["sizes"]):
            assert [float(t) for t in x[i, off:off + 6]] == v["features"][n]
            assert abs(float(x[i, off:off + s].astype(np.float64).sum()) - v["feature_sum"][n]) < 1e-9
            off += s


def test_protobuf_golden_is_reproducible(golden_dir, tmp_path):
    """The committed bytes are what the installed protobuf runtime produces today (skipped when protobuf is absent)."""
    pytest.importorskip("google.protobuf")
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("make_tfrecord_golden", os.path.join(golden_dir, "make_tfrecord_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Example, SequenceExample = mod.build_messages()
    ex = SequenceExample()
    ex.context.feature["labels"].int64_list.value.extend([1, 300])
    ex.feature_lists.feature_list["rgb"].feature.add().bytes_list.value.append(b"\\x01\\x02")
    # the oracle's encoder and protobuf agree byte for byte on a record whose maps have a single key each
    ctx = {"labels": tr.int64_feature([1, 300])}
    assert tr.sequence_example(ctx, {"rgb": [tr.bytes_feature([b"\\x01\\x02"])]}) == ex.SerializeToString(deterministic=True)
