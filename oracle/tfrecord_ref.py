"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- pure-Python restatement of the TFRecord container and the
tf.train.Example / SequenceExample wire format used by the YouTube-8M readers (W/readers.py:94-125,189-259), plus a
WRITER so tests can fabricate shards (the reference ships no data files).

Pinned only as far as public specifications go: CRC-32C against the RFC 3720 B.4 check values; the protobuf encoding
follows the published tensorflow/core/example/{example,feature}.proto field numbers:
  Example{features=1} SequenceExample{context=1, feature_lists=2} Features{map feature=1} FeatureLists{map feature_list=1}
  FeatureList{repeated Feature feature=1} Feature{bytes_list=1, float_list=2, int64_list=3} *List{repeated value=1}.
"""
import struct

import numpy as np

_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ _POLY if _c & 1 else _c >> 1
    _TABLE.append(_c)


def crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c = (c >> 8) ^ _TABLE[(c ^ b) & 0xFF]
    return c ^ 0xFFFFFFFF


def masked_crc(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def bytes_feature(values):
    return _ld(1, b"".join(_ld(1, v) for v in values))


def float_feature(values, packed=True):
    vals = np.asarray(values, dtype="<f4")
    if packed:
        return _ld(2, _ld(1, vals.tobytes()))
    return _ld(2, b"".join(_varint((1 << 3) | 5) + struct.pack("<f", float(v)) for v in vals))


def int64_feature(values, packed=True):
    if packed:
        return _ld(3, _ld(1, b"".join(_varint(int(v)) for v in values)))
    return _ld(3, b"".join(_varint((1 << 3) | 0) + _varint(int(v)) for v in values))


def _features(d):
    return b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, v)) for k, v in d.items())


def example(features):
    return _ld(1, _features(features))


def sequence_example(context, feature_lists):
    fl = b"".join(_ld(1, _ld(1, k.encode()) + _ld(2, b"".join(_ld(1, f) for f in frames))) for k, frames in feature_lists.items())
    return _ld(1, _features(context)) + _ld(2, fl)


def record(payload):
    hdr = struct.pack("<Q", len(payload))
    return hdr + struct.pack("<I", masked_crc(hdr)) + payload + struct.pack("<I", masked_crc(payload))


def write_frame_shard(path, videos, feature_names, packed=True):
    """videos: list of dict(video_id=bytes, labels=[int], frames={name: uint8 [n, size]})."""
    with open(path, "wb") as fh:
        for v in videos:
            ctx = {"video_id": bytes_feature([v["video_id"]]), "labels": int64_feature(v["labels"], packed)}
            fls = {n: [bytes_feature([np.ascontiguousarray(row, dtype=np.uint8).tobytes()]) for row in v["frames"][n]] for n in feature_names}
            fh.write(record(sequence_example(ctx, fls)))


def write_video_shard(path, videos, feature_names, packed=True):
    """videos: list of dict(video_id=bytes, labels=[int], features={name: float32 [size]})."""
    with open(path, "wb") as fh:
        for v in videos:
            feats = {"video_id": bytes_feature([v["video_id"]]), "labels": int64_feature(v["labels"], packed)}
            for n in feature_names:
                feats[n] = float_feature(v["features"][n], packed)
            fh.write(record(example(feats)))


def expected_frame_batch(videos, feature_names, feature_sizes, max_frames, num_classes):
    """What YT8MFrameFeatureReader must deliver BEFORE dequantisation (W/readers.py:159-187,217-250)."""
    D = sum(feature_sizes)
    q = np.zeros((len(videos), max_frames, D), dtype=np.uint8)
    nf = np.zeros(len(videos), dtype=np.int32)
    lab = np.zeros((len(videos), num_classes), dtype=bool)
    for i, v in enumerate(videos):
        off = 0
        for n, s in zip(feature_names, feature_sizes):
            fr = np.asarray(v["frames"][n], dtype=np.uint8).reshape(-1, s)
            k = min(len(fr), max_frames)
            q[i, :k, off:off + s] = fr[:k]
            nf[i] = k
            off += s
        for l in v["labels"]:
            lab[i, l] = True
    return q, nf, lab
