"""Input readers with the reference's class names / constructor arguments (W/readers.py:66-259) on top of the native
TFRecord + protobuf decoder in libyt8m_hip.so (csrc/tfrecord.hip).  Instead of TF queue runners, ``prepare_reader``
returns a Python iterator of batches; frame features are delivered as RAW uint8 [B, max_frames, D] (pinned host
memory -> device), the fused device transform (ops.dequant_l2norm) does Dequantize + padding + L2-normalise.
"""
import ctypes
import glob

import torch

from . import _lib
from .flags import DEFINE_integer

DEFINE_integer("num_readers", 8, "How many threads to use for reading input files.")
ID_STRIDE = 32


class BaseReader(object):
    """W/readers.py:58-63."""

    def prepare_reader(self, unused_filename_queue):
        raise NotImplementedError()


def _files(pattern_or_list):
    if isinstance(pattern_or_list, (list, tuple)):
        files = list(pattern_or_list)
    else:
        files = sorted(glob.glob(pattern_or_list))
    if not files:
        raise IOError("Unable to find training files. data_pattern='%s'." % (pattern_or_list,))   # W/train.py:193-195
    return files


def _c_names(names, sizes):
    arr = (ctypes.c_char_p * len(names))(*[n.encode() for n in names])
    szs = (ctypes.c_int32 * len(sizes))(*sizes)
    return arr, szs


def _ids(buf, n):
    raw = buf[:n].numpy().tobytes()
    return [raw[i * ID_STRIDE:(i + 1) * ID_STRIDE].split(b"\0", 1)[0] for i in range(n)]


class _Open(object):
    def __init__(self, path, check_crc):
        self.h = ctypes.c_void_p()
        _lib.check(_lib.lib().yt8m_tfrecord_open(path.encode(), int(check_crc), ctypes.byref(self.h)))

    def close(self):
        if self.h:
            _lib.lib().yt8m_tfrecord_close(self.h)
            self.h = None


def _view(ptr, shape, dtype):
    """Zero-copy torch view of a host buffer lent by the native prefetcher (valid until the next acquire)."""
    import numpy
    n = 1
    for d in shape:
        n *= d
    if n == 0:
        return torch.empty(shape, dtype=dtype)
    ct = {torch.uint8: ctypes.c_uint8, torch.int32: ctypes.c_int32, torch.float32: ctypes.c_float}[dtype]
    arr = numpy.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ct)), shape=(n,))
    return torch.from_numpy(arr).view(*shape)


def _prefetched(reader, filenames, batch_size, device, check_crc, num_threads, queue_depth, frame_level, copy=True):
    """Batches from the native multi-threaded shard prefetcher (yt8m_prefetch_*): the reference's num_readers reader threads
    (W/train.py:199-209).  Same tuples as the sequential path; batches never span shards."""
    files = _files(filenames)
    names, sizes = _c_names(reader.feature_names, reader.feature_sizes)
    paths = (ctypes.c_char_p * len(files))(*[f.encode() for f in files])
    D = sum(reader.feature_sizes)
    L = _lib.lib()
    h = ctypes.c_void_p()
    _lib.check(L.yt8m_prefetch_open(paths, len(files), int(frame_level), names, sizes, len(reader.feature_names),
                                    getattr(reader, "max_frames", 1), reader.num_classes, batch_size, int(num_threads),
                                    int(queue_depth), int(check_crc), ctypes.byref(h)))
    copy_stream = None
    try:
        data, nfp, labp, idp = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        stride, n, pinned = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int(0)
        while True:
            _lib.check(L.yt8m_prefetch_acquire(h, ctypes.byref(data), ctypes.byref(nfp), ctypes.byref(labp), ctypes.byref(idp),
                                               ctypes.byref(stride), ctypes.byref(n), ctypes.byref(pinned)))
            k = n.value
            if k == 0:
                break
            ids = _view(idp.value, (k, stride.value), torch.uint8)
            raw = ids.numpy().tobytes()
            vids = [raw[i * stride.value:(i + 1) * stride.value].split(b"\0", 1)[0] for i in range(k)]
            lab = _view(labp.value, (k, reader.num_classes), torch.uint8)
            if frame_level:
                feat = _view(data.value, (k, reader.max_frames, D), torch.uint8)
                extra = _view(nfp.value, (k,), torch.int32)
            else:
                feat = _view(data.value, (k, D), torch.float32)
                extra = None
            if device is not None:
                # The slot is lent: its copies must finish before the next acquire.  They run on a copy stream of their own, and only
                # THAT stream is waited for -- synchronising the consumer's stream here would also wait for the training step it has
                # queued, i.e. serialise the PCIe transfer with the step instead of hiding it under it (tools/reader_bench.py).
                cur = torch.cuda.current_stream(device)
                if copy_stream is None:
                    copy_stream = torch.cuda.Stream(device=device)
                with torch.cuda.stream(copy_stream):
                    feat, lab = feat.to(device, non_blocking=True), lab.to(device, non_blocking=True).bool()
                    extra = extra.to(device, non_blocking=True) if extra is not None else torch.ones(k, device=device)
                copy_stream.synchronize()
                for t in (feat, lab, extra):
                    t.record_stream(cur)                 # allocated on the copy stream, consumed on the caller's
            elif copy:
                feat, lab = feat.clone(), lab.bool()
                extra = extra.clone() if extra is not None else torch.ones(k)
            else:                                        # zero-copy views of the lent slot: valid until the next batch is requested
                extra = extra if extra is not None else torch.ones(k)
            yield vids, feat, lab, extra
    finally:
        L.yt8m_prefetch_close(h)


class YT8MAggregatedFeatureReader(BaseReader):
    """Video-level Examples: sparse int64 'labels', 'video_id', fixed-length float features (W/readers.py:66-125)."""

    def __init__(self, num_classes=4716, feature_sizes=[1024], feature_names=["mean_inc3"]):
        assert len(feature_names) == len(feature_sizes), \
            "length of feature_names (={}) != length of feature_sizes (={})".format(len(feature_names), len(feature_sizes))
        self.num_classes = num_classes
        self.feature_sizes = list(feature_sizes)
        self.feature_names = list(feature_names)

    def prepare_reader(self, filenames, batch_size=1024, device=None, check_crc=True, num_threads=0, queue_depth=4):
        """Yields (video_ids, features float32 [n, D], labels bool [n, num_classes], ones [n]) with n <= batch_size
        (the last batch of a file may be smaller: read_up_to semantics, W/readers.py:104).  num_threads > 0: the native
        multi-threaded shard prefetcher (--num_readers) instead of the sequential loop."""
        assert len(self.feature_names) > 0, "self.feature_names is empty!"
        if num_threads > 0:
            yield from _prefetched(self, filenames, batch_size, device, check_crc, num_threads, queue_depth, False)
            return
        names, sizes = _c_names(self.feature_names, self.feature_sizes)
        D = sum(self.feature_sizes)
        pin = torch.cuda.is_available()
        x = torch.empty((batch_size, D), dtype=torch.float32, pin_memory=pin)
        lab = torch.empty((batch_size, self.num_classes), dtype=torch.uint8, pin_memory=pin)
        ids = torch.empty((batch_size, ID_STRIDE), dtype=torch.uint8)
        n = ctypes.c_int64(0)
        L = _lib.lib()
        for path in _files(filenames):
            rd = _Open(path, check_crc)
            try:
                while True:
                    _lib.check(L.yt8m_tfrecord_read_video_batch(rd.h, names, sizes, len(self.feature_names), self.num_classes,
                                                                batch_size, x.data_ptr(), lab.data_ptr(), ids.data_ptr(),
                                                                ID_STRIDE, ctypes.byref(n)))
                    if n.value == 0:
                        break
                    k = n.value
                    xb, lb = x[:k], lab[:k].bool()
                    if device is not None:
                        xb, lb = xb.to(device, non_blocking=True), lb.to(device, non_blocking=True)
                        if pin:
                            torch.cuda.current_stream().synchronize()      # the staging buffers are re-used
                    else:
                        xb, lb = xb.clone(), lb.clone()
                    yield _ids(ids, k), xb, lb, torch.ones(k, device=xb.device)
                    if k < batch_size:
                        break
            finally:
                rd.close()


class YT8MFrameFeatureReader(BaseReader):
    """Frame-level SequenceExamples: 'labels' / 'video_id' context, one uint8 bytes feature per frame and feature name
    (W/readers.py:127-259)."""

    def __init__(self, num_classes=4716, feature_sizes=[1024], feature_names=["inc3"], max_frames=300):
        assert len(feature_names) == len(feature_sizes), \
            "length of feature_names (={}) != length of feature_sizes (={})".format(len(feature_names), len(feature_sizes))
        self.num_classes = num_classes
        self.feature_sizes = list(feature_sizes)
        self.feature_names = list(feature_names)
        self.max_frames = max_frames

    def prepare_reader(self, filenames, batch_size=128, device=None, check_crc=True, num_threads=0, queue_depth=4, copy=True):
        """Yields (video_ids, q uint8 [n, max_frames, D], labels bool [n, num_classes], num_frames int32 [n]).
        num_threads > 0: the native multi-threaded shard prefetcher (--num_readers); with device=None and copy=False its batches
        are zero-copy views of the prefetcher's pinned slot (labels stay uint8), valid until the next batch is requested."""
        assert len(self.feature_names) > 0, "No feature selected: feature_names is empty!"
        if num_threads > 0:
            yield from _prefetched(self, filenames, batch_size, device, check_crc, num_threads, queue_depth, True, copy)
            return
        names, sizes = _c_names(self.feature_names, self.feature_sizes)
        D = sum(self.feature_sizes)
        pin = torch.cuda.is_available()
        q = torch.empty((batch_size, self.max_frames, D), dtype=torch.uint8, pin_memory=pin)
        nf = torch.empty((batch_size,), dtype=torch.int32, pin_memory=pin)
        lab = torch.empty((batch_size, self.num_classes), dtype=torch.uint8, pin_memory=pin)
        ids = torch.empty((batch_size, ID_STRIDE), dtype=torch.uint8)
        n = ctypes.c_int64(0)
        L = _lib.lib()
        for path in _files(filenames):
            rd = _Open(path, check_crc)
            try:
                while True:
                    _lib.check(L.yt8m_tfrecord_read_frame_batch(rd.h, names, sizes, len(self.feature_names), self.max_frames,
                                                                self.num_classes, batch_size, q.data_ptr(), nf.data_ptr(),
                                                                lab.data_ptr(), ids.data_ptr(), ID_STRIDE, ctypes.byref(n)))
                    if n.value == 0:
                        break
                    k = n.value
                    qb, nb, lb = q[:k], nf[:k], lab[:k].bool()
                    if device is not None:
                        qb, nb, lb = (t.to(device, non_blocking=True) for t in (qb, nb, lb))
                        if pin:
                            torch.cuda.current_stream().synchronize()
                    else:
                        qb, nb, lb = qb.clone(), nb.clone(), lb.clone()
                    yield _ids(ids, k), qb, lb, nb
                    if k < batch_size:
                        break
            finally:
                rd.close()
