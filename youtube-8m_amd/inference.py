"""Prediction export helpers: the CSV line format of W/inference.py:76-89,163 with the
per-video top-k selection done on the MI355X (yt8m_topk_rows) so that only B*k (class, score) pairs cross PCIe."""
import ctypes

import numpy

from . import _lib, ops

CSV_HEADER = "VideoId,LabelConfidencePairs\n"          # W/inference.py:163


def format_lines(video_ids, predictions, top_k):
    """Yields "<video_id>,<class> <score> <class> <score> ...\\n" with the top_k classes sorted by descending score,
    each pair printed with "%i %f" (W/inference.py:76-89).  predictions: device tensor [B,V] (device top-k) or a numpy
    array (host path, identical output on tie-free scores)."""
    if hasattr(predictions, "is_cuda") and predictions.is_cuda:
        vals, idx = ops.topk_rows(predictions, top_k)
        vals, idx = vals.cpu().numpy(), idx.cpu().numpy()
        for v in range(len(video_ids)):
            vid = video_ids[v].decode("utf-8") if isinstance(video_ids[v], bytes) else str(video_ids[v])
            yield vid + "," + " ".join("%i %f" % (int(c), float(s)) for c, s in zip(idx[v], vals[v])) + "\n"
        return
    import numpy
    for v in range(len(video_ids)):
        top = numpy.argpartition(predictions[v], -top_k)[-top_k:]
        line = sorted(((int(c), float(predictions[v][c])) for c in top), key=lambda p: -p[1])
        vid = video_ids[v].decode("utf-8") if isinstance(video_ids[v], bytes) else str(video_ids[v])
        yield vid + "," + " ".join("%i %f" % pair for pair in line) + "\n"


def write_to_record(path, id_batch, label_batch, predictions, feature_name="predictions"):
    """W/inference-pre-ensemble.py:291-308 (write_to_record / get_output_feature): dumps one tf.train.Example per video --
    {"video_id", "labels" = nonzero(label row), feature_name = prediction row} -- into a TFRecord file, the input format of
    the ensemble stage.  Host arrays (device tensors are copied once); the record framing / protobuf encoding is native
    (yt8m_tfrecord_write_predictions).  Files are named predictions-%04d.tfrecord by the caller, as in the reference."""
    if hasattr(predictions, "is_cuda"):
        predictions = predictions.detach().to("cpu").numpy()
    if hasattr(label_batch, "is_cuda"):
        label_batch = label_batch.detach().to("cpu").numpy()
    pred = numpy.ascontiguousarray(predictions, dtype=numpy.float32)
    lab = numpy.ascontiguousarray(numpy.asarray(label_batch) != 0, dtype=numpy.uint8)
    n, V = pred.shape
    assert lab.shape == (n, V) and len(id_batch) == n
    ids = [v if isinstance(v, bytes) else str(v).encode("utf-8") for v in id_batch]
    stride = max([len(v) for v in ids] + [1]) + 1
    buf = numpy.zeros((n, stride), dtype=numpy.uint8)
    for i, v in enumerate(ids):
        buf[i, :len(v)] = numpy.frombuffer(v, dtype=numpy.uint8)
    _lib.check(_lib.lib().yt8m_tfrecord_write_predictions(str(path).encode(), n, buf.ctypes.data_as(ctypes.c_void_p), stride,
                                                          lab.ctypes.data_as(ctypes.c_void_p),
                                                          pred.ctypes.data_as(ctypes.c_void_p), V, feature_name.encode()))
