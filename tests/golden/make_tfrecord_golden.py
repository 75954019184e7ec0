"""Generates tests/golden/tfrecord_{frame,video}.bin + tfrecord_golden.json: tf.train.SequenceExample / Example records encoded by
the installed protobuf runtime (google.protobuf, NOT this repository's encoder) from descriptors built here after the published
tensorflow/core/example/{feature,example}.proto (field numbers and packing as released with TF 1.0), framed as TFRecords.

The native reader (csrc/tfrecord.hip) and the oracle (oracle/tfrecord_ref.py) are then pinned against bytes that an
independent implementation of the wire format produced.  Run in the build container:
    python tests/golden/make_tfrecord_golden.py
"""
import json
import os
import struct
import sys

import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import tfrecord_ref as tr  # noqa: E402  (only for the CRC-32C of the record framing, pinned on RFC 3720 vectors)

T = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=T.LABEL_OPTIONAL, type_name=None, packed=None, oneof=None):
    f = msg.field.add()
    f.name, f.number, f.type, f.label = name, number, ftype, label
    if type_name:
        f.type_name = type_name
    if packed is not None:
        f.options.packed = packed
    if oneof is not None:
        f.oneof_index = oneof
    return f


def _map_entry(parent, entry_name, value_type_name):
    e = parent.nested_type.add()
    e.name = entry_name
    e.options.map_entry = True
    _field(e, "key", 1, T.TYPE_STRING)
    _field(e, "value", 2, T.TYPE_MESSAGE, type_name=value_type_name)


def build_messages():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name, fd.package, fd.syntax = "yt8m_golden/example.proto", "tensorflow", "proto3"
    m = fd.message_type.add(); m.name = "BytesList"; _field(m, "value", 1, T.TYPE_BYTES, T.LABEL_REPEATED)
    m = fd.message_type.add(); m.name = "FloatList"; _field(m, "value", 1, T.TYPE_FLOAT, T.LABEL_REPEATED, packed=True)
    m = fd.message_type.add(); m.name = "Int64List"; _field(m, "value", 1, T.TYPE_INT64, T.LABEL_REPEATED, packed=True)
    m = fd.message_type.add(); m.name = "Feature"
    m.oneof_decl.add().name = "kind"
    _field(m, "bytes_list", 1, T.TYPE_MESSAGE, type_name=".tensorflow.BytesList", oneof=0)
    _field(m, "float_list", 2, T.TYPE_MESSAGE, type_name=".tensorflow.FloatList", oneof=0)
    _field(m, "int64_list", 3, T.TYPE_MESSAGE, type_name=".tensorflow.Int64List", oneof=0)
    m = fd.message_type.add(); m.name = "Features"
    _map_entry(m, "FeatureEntry", ".tensorflow.Feature")
    _field(m, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".tensorflow.Features.FeatureEntry")
    m = fd.message_type.add(); m.name = "FeatureList"
    _field(m, "feature", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".tensorflow.Feature")
    m = fd.message_type.add(); m.name = "FeatureLists"
    _map_entry(m, "FeatureListEntry", ".tensorflow.FeatureList")
    _field(m, "feature_list", 1, T.TYPE_MESSAGE, T.LABEL_REPEATED, type_name=".tensorflow.FeatureLists.FeatureListEntry")
    m = fd.message_type.add(); m.name = "Example"
    _field(m, "features", 1, T.TYPE_MESSAGE, type_name=".tensorflow.Features")
    m = fd.message_type.add(); m.name = "SequenceExample"
    _field(m, "context", 1, T.TYPE_MESSAGE, type_name=".tensorflow.Features")
    _field(m, "feature_lists", 2, T.TYPE_MESSAGE, type_name=".tensorflow.FeatureLists")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("tensorflow." + n))
    return get("Example"), get("SequenceExample")


def frame_record(SequenceExample, vid):
    ex = SequenceExample()
    ex.context.feature["video_id"].bytes_list.value.append(vid["video_id"])
    ex.context.feature["labels"].int64_list.value.extend(vid["labels"])
    for name, arr in vid["frames"].items():
        fl = ex.feature_lists.feature_list[name]
        for row in arr:
            fl.feature.add().bytes_list.value.append(row.tobytes())
    return ex.SerializeToString(deterministic=True)


def video_record(Example, vid):
    ex = Example()
    ex.features.feature["video_id"].bytes_list.value.append(vid["video_id"])
    ex.features.feature["labels"].int64_list.value.extend(vid["labels"])
    for name, arr in vid["features"].items():
        ex.features.feature[name].float_list.value.extend([float(v) for v in arr])
    return ex.SerializeToString(deterministic=True)


def framed(payload):
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", tr.masked_crc(head)) + payload + struct.pack("<I", tr.masked_crc(payload))


def main():
    Example, SequenceExample = build_messages()
    rs = np.random.RandomState(2017)
    names, sizes = ["rgb", "audio"], [1024, 128]
    vids = []
    for i, (k, labels) in enumerate([(5, [1, 4715, 17]), (1, []), (12, [300]), (9, [2, 2, 9])]):
        vids.append(dict(video_id=("gold%03d" % i).encode(), labels=labels,
                         frames={n: rs.randint(0, 256, size=(k, s)).astype(np.uint8) for n, s in zip(names, sizes)},
                         features={"mean_" + n: rs.randn(s).astype(np.float32) for n, s in zip(names, sizes)}))
    with open(os.path.join(HERE, "tfrecord_frame.bin"), "wb") as f:
        for v in vids:
            f.write(framed(frame_record(SequenceExample, v)))
    with open(os.path.join(HERE, "tfrecord_video.bin"), "wb") as f:
        for v in vids:
            f.write(framed(video_record(Example, v)))
    meta = dict(names=names, sizes=sizes, max_frames=10, num_classes=4716, videos=[
        dict(video_id=v["video_id"].decode(), labels=v["labels"], num_frames=int(v["frames"]["rgb"].shape[0]),
             frame_checksum={n: int(a.astype(np.uint64).sum()) for n, a in v["frames"].items()},
             frame_first_row={n: a[0, :8].tolist() for n, a in v["frames"].items()},
             frame_last_row={n: a[min(a.shape[0], 10) - 1, -8:].tolist() for n, a in v["frames"].items()},
             features={n: [float(np.float32(x)) for x in a[:6]] for n, a in v["features"].items()},
             feature_sum={n: float(np.float64(a.astype(np.float64).sum())) for n, a in v["features"].items()})
        for v in vids], protobuf_version=__import__("google.protobuf").protobuf.__version__)
    with open(os.path.join(HERE, "tfrecord_golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", [os.path.getsize(os.path.join(HERE, n)) for n in ("tfrecord_frame.bin", "tfrecord_video.bin")])


if __name__ == "__main__":
    main()
