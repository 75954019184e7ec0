"""Prediction export helpers: the CSV line format of W/inference.py:76-89,163 with the
per-video top-k selection done on the MI355X (yt8m_topk_rows) so that only B*k (class, score) pairs cross PCIe."""
from . import ops

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
