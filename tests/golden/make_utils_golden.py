"""Generates tests/golden/utils_kat.json by IMPORTING the reference's own W/utils.py and W/inference.py (W =
/root/reference/youtube-8m-wangheda) and calling the functions of theirs that are plain Python / numpy:

    utils.Dequantize                      W/utils.py:23-38     (the a2 row of SURVEY.md section 8: the only non-metric hot-path
                                                               function the reference itself can vouch for in this container)
    utils.GetListOfFeatureNamesAndSizes   W/utils.py:140-161
    inference.format_lines                W/inference.py:76-89 (the CSV line format of the f2 row)

Both modules `import tensorflow` at the top; none of the three functions touches it.  The import is satisfied by EMPTY module
objects (no TensorFlow behaviour is stood in for: the only attributes given are the names the import statements themselves bind --
tf.logging for utils.py:20, tf.app / flags / gfile / logging for inference.py:23-26 -- and W/losses.py / W/readers.py, which
inference.py imports and which do not parse under Python 3, are likewise empty).  Only numbers and strings are written; no
reference source is stored.  Run in the build container only (needs /root/reference; never runs on the GPU box):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_utils_golden.py
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/youtube-8m-wangheda"


class _Log(object):                       # utils.py:155-158 calls logging.error on a length mismatch: record the call, nothing else
    calls = []

    def error(self, msg, *a):
        _Log.calls.append(str(msg))


tf = types.ModuleType("tensorflow")
tf.logging = _Log()
for name in ("app", "flags", "gfile"):
    setattr(tf, name, types.ModuleType("tensorflow." + name))
tf.flags.FLAGS = None
sys.modules["tensorflow"] = tf
for n in ["tensorflow.python", "tensorflow.python.platform", "tensorflow.python.platform.gfile"]:
    sys.modules[n] = types.ModuleType(n)
sys.modules["tensorflow.python.platform"].gfile = sys.modules["tensorflow.python.platform.gfile"]
for n in ("losses", "readers"):          # imported by inference.py:29-30; Python-2 syntax / TF graph code, unused by format_lines
    sys.modules[n] = types.ModuleType(n)
sys.path.insert(0, REF)
import utils  # noqa: E402
import inference  # noqa: E402

out = {"_generator": "tests/golden/make_utils_golden.py", "_python": sys.version.split()[0], "_numpy": np.__version__}

# --- Dequantize: all 256 byte values, as float32 (the reader casts uint8 -> float32 first: W/readers.py:178-185) ------------
q = np.arange(256, dtype=np.float32)
d = utils.Dequantize(q)
assert d.dtype == np.float32
out["dequantize_default"] = {"dtype": str(d.dtype), "values": [float(v) for v in d], "hex": [np.float32(v).tobytes().hex() for v in d]}
d2 = utils.Dequantize(q, 4, -1)
out["dequantize_max4_min-1"] = {"values": [float(v) for v in d2], "hex": [np.float32(v).tobytes().hex() for v in d2]}
d64 = utils.Dequantize(np.arange(256, dtype=np.float64))
out["dequantize_default_f64"] = [float(v) for v in d64]
try:
    utils.Dequantize(q, 1, 1)
    out["dequantize_bad_range"] = "no error"
except AssertionError:
    out["dequantize_bad_range"] = "AssertionError"

# --- GetListOfFeatureNamesAndSizes ------------------------------------------------------------------------------------------
cases = []
for names, sizes in [("mean_rgb", "1024"), ("rgb, audio", "1024, 128"), (" rgb ,audio", "1024,128"), ("mean_rgb,mean_audio", "1024")]:
    _Log.calls = []
    try:
        r = utils.GetListOfFeatureNamesAndSizes(names, sizes)
        cases.append({"names": names, "sizes": sizes, "result": [list(r[0]), list(r[1])], "logged_errors": len(_Log.calls)})
    except Exception as e:  # noqa: BLE001
        cases.append({"names": names, "sizes": sizes, "raises": type(e).__name__})
try:
    utils.GetListOfFeatureNamesAndSizes("rgb", "10x")
except Exception as e:  # noqa: BLE001
    cases.append({"names": "rgb", "sizes": "10x", "raises": type(e).__name__})
out["feature_names_and_sizes"] = cases

# --- inference.format_lines --------------------------------------------------------------------------------------------------
fl = []
for seed, (B, V, k) in enumerate([(4, 4716, 20), (3, 25, 20), (2, 20, 20), (5, 100, 1)]):
    rs = np.random.RandomState(500 + seed)
    p = np.stack([(rs.permutation(V) + rs.rand()) / V for _ in range(B)]).astype(np.float32)      # tie-free rows (checked below)
    assert all(len(set(row.tolist())) == V for row in p)
    ids = [("vid%04d" % (seed * 10 + i)).encode("utf-8") for i in range(B)]
    fl.append({"seed": 500 + seed, "B": B, "V": V, "top_k": k, "lines": list(inference.format_lines(ids, p, k))})
out["format_lines"] = fl

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "utils_kat.json")
with open(dst, "w") as fh:
    json.dump(out, fh, indent=0, sort_keys=True)
print("wrote", dst, "Dequantize(0) =", out["dequantize_default"]["values"][0], "Dequantize(255) =", out["dequantize_default"]["values"][255])
print(cases)
print(fl[1]["lines"][0][:80])
