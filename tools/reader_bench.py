"""Input-pipeline throughput: synthetic TFRecord shards -> native multi-threaded prefetcher
(yt8m_prefetch_*, the reference's --num_readers reader threads: W/readers.py:189-259, W/train.py:199-209) -> pinned host slot ->
H2D -> TrainGraph.step, for BASELINE configs[3] (LstmModel, B = 128) and configs[2] (NetVLADModel, B = 1024).

Reports, per configuration and reader thread count:
  decode   videos/s the prefetcher alone delivers into pinned host memory (no GPU work)
  feed     videos/s of decode + H2D copy (no training step)
  fed      videos/s of the training loop fed by the reader
  resident videos/s of the same training step on a batch already in HBM (what bench.py's `value` measures)
and the PCIe rate of the copies.  A configuration is reader / PCIe-bound when `fed` < `resident`.
    python tools/reader_bench.py [--videos 2048] [--threads 4,8,16,32] [--steps 24] > profiles/r4_reader_bench.txt"""
import argparse
import ctypes
import os
import struct
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import yt8m_amd._lib as L  # noqa: E402
import yt8m_amd.frame_level_models as flm  # noqa: E402
import yt8m_amd.readers as readers  # noqa: E402
import yt8m_amd.train as train  # noqa: E402
from yt8m_amd.flags import FLAGS  # noqa: E402
from yt8m_amd.variables import reset_default_graph  # noqa: E402

F, D_RGB, D_AUDIO, V = 300, 1024, 128, 4716


# ---- a minimal tf.train.SequenceExample / TFRecord writer (wire format: feature.proto / example.proto; framing: u64 length,
# masked crc32c of the length, payload, masked crc32c of the payload).  CRCs come from the library's own yt8m_crc32c_masked.
def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _bytes_feature(values):
    return _ld(1, b"".join(_ld(1, v) for v in values))                      # Feature.bytes_list = 1, BytesList.value = 1


def _int64_feature(values):
    return _ld(3, _ld(1, b"".join(_varint(v) for v in values)))              # Feature.int64_list = 3, packed Int64List.value = 1


def _map_entry(key, value):
    return _ld(1, _ld(1, key.encode()) + _ld(2, value))                      # map<string, X> entry: key = 1, value = 2


def sequence_example(video_id, labels, rgb, audio):
    ctx = _map_entry("video_id", _bytes_feature([video_id])) + _map_entry("labels", _int64_feature(labels))
    fl = b""
    for name, mat in (("rgb", rgb), ("audio", audio)):
        feats = b"".join(_ld(1, _bytes_feature([row.tobytes()])) for row in mat)       # FeatureList.feature = 1
        fl += _map_entry(name, feats)
    return _ld(1, ctx) + _ld(2, fl)                                           # SequenceExample.context = 1, feature_lists = 2


def record(payload, lib):
    hdr = struct.pack("<Q", len(payload))
    return hdr + struct.pack("<I", lib.yt8m_crc32c_masked(hdr, 8)) + payload + struct.pack("<I", lib.yt8m_crc32c_masked(payload, len(payload)))


def write_shards(dirname, videos, shards, lib, distinct=16):
    """`videos` videos of F frames over `shards` files; `distinct` different payloads repeated (the decoder's work does not depend on
    the bytes; writing 2048 distinct videos from Python would take minutes)."""
    rs = np.random.RandomState(0)
    recs = []
    for i in range(distinct):
        rgb = rs.randint(0, 256, size=(F, D_RGB), dtype=np.uint8)
        audio = rs.randint(0, 256, size=(F, D_AUDIO), dtype=np.uint8)
        labels = sorted(set(int(x) for x in rs.randint(0, V, size=3)))
        recs.append(record(sequence_example(b"vid%05d" % i, labels, rgb, audio), lib))
    paths, per = [], videos // shards
    for s in range(shards):
        p = os.path.join(dirname, "train%03d.tfrecord" % s)
        with open(p, "wb") as f:
            for k in range(per):
                f.write(recs[(s * per + k) % distinct])
        paths.append(p)
    return paths, per * shards, sum(len(r) for r in recs) / float(distinct)


def build(model, B, dev):
    FLAGS.reset()
    g = reset_default_graph(device=dev, seed=0)
    return train.TrainGraph(model, batch_size=B, graph=g)


def run_config(name, model_cls, B, paths, nvid, rec_bytes, threads, steps, dev):
    rd = readers.YT8MFrameFeatureReader(num_classes=V, feature_sizes=[D_RGB, D_AUDIO], feature_names=["rgb", "audio"], max_frames=F)
    bytes_per_video = F * (D_RGB + D_AUDIO) + V + 4
    print("== %s: B = %d, %d videos in %d shards (%.1f KB per record), %.1f KB per video over PCIe" % (
        name, B, nvid, len(paths), rec_bytes / 1e3, bytes_per_video / 1e3))
    tg = build(model_cls(), B, dev)
    # resident: the same step on a batch already in HBM
    it = rd.prepare_reader(paths, batch_size=B, device=dev, check_crc=False, num_threads=4)
    _, q0, y0, nf0 = next(it)
    it.close()
    assert q0.shape[0] == B, "short batch %d: shards hold fewer than B videos" % q0.shape[0]
    for _ in range(3):
        tg.step(q0, y0, nf0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tg.step(q0, y0, nf0)
    torch.cuda.synchronize()
    resident = steps * B / (time.perf_counter() - t0)
    print("   resident (inputs in HBM): %9.0f videos/s  (%.2f ms/step)" % (resident, B / resident * 1e3))
    paths = list(paths) * (4 if B >= 1024 else 2)                # several passes over the shards: steady state, not the pre-filled queue
    for crc in (False, True):
        for nt in threads:
            res = {}
            for mode in ("decode", "feed", "fed"):
                it = rd.prepare_reader(paths, batch_size=B, device=None if mode == "decode" else dev, check_crc=crc, num_threads=nt,
                                       queue_depth=nt + 2, copy=False)      # a worker decodes a whole batch into one slot: slots >= threads
                n, t0 = 0, None
                for _, q, y, nf in it:
                    if t0 is None:                                   # the first batch pays thread start-up and the first page faults
                        t0 = time.perf_counter()
                        continue
                    if mode == "fed":
                        tg.step(q, y, nf)
                    n += q.shape[0]
                if mode != "decode":
                    torch.cuda.synchronize()
                el = time.perf_counter() - t0
                try:
                    it.close()
                except Exception:
                    pass
                res[mode] = n / el
            print("   crc %-5s threads %2d: decode %8.0f  feed (decode + H2D) %8.0f  fed training %8.0f videos/s   H2D at feed %.1f GB/s, at fed %.1f GB/s%s"
                  % (crc, nt, res["decode"], res["feed"], res["fed"], res["feed"] * bytes_per_video / 1e9,
                     res["fed"] * bytes_per_video / 1e9, "   <-- reader / PCIe bound" if res["fed"] < 0.95 * resident else ""))
    del tg
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=8192)
    ap.add_argument("--shards", type=int, default=8, help="batches never span shards: videos / shards must be >= the largest batch (1024)")
    ap.add_argument("--threads", default="4,8,16")
    ap.add_argument("--steps", type=int, default=24)
    a = ap.parse_args()
    lib = L.lib()
    lib.yt8m_crc32c_masked.argtypes = [ctypes.c_char_p, ctypes.c_int64]
    lib.yt8m_crc32c_masked.restype = ctypes.c_uint32
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    threads = [int(t) for t in a.threads.split(",")]
    print("host: %d usable cores; device: %s" % (len(os.sched_getaffinity(0)), torch.cuda.get_device_name(0)))
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        t0 = time.perf_counter()
        assert a.videos // a.shards >= 1024, "a shard must hold at least one configs[2] batch of 1024 videos"
        paths, nvid, rec_bytes = write_shards(d, a.videos, a.shards, lib)
        print("wrote %d videos (%.2f GB) into %s in %.1f s (page-cache resident: this measures decode + copy, not the disks)" % (
            nvid, nvid * rec_bytes / 1e9, d, time.perf_counter() - t0))
        run_config("BASELINE configs[3] LstmModel", flm.LstmModel, 128, paths, nvid, rec_bytes, threads, a.steps, dev)
        run_config("BASELINE configs[2] NetVLADModel", flm.NetVLADModel, 1024, paths, nvid, rec_bytes, threads, max(3, a.steps // 4), dev)


if __name__ == "__main__":
    main()
