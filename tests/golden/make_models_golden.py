"""Writes tests/golden/models_kat.json: expected outputs of the seeded tiny cases of tests/golden/model_cases.py, computed with the
fp64 numpy restatement (oracle/np_ref.py).  The reference itself cannot produce these (Python 2 / TensorFlow 1.0):
the fixture pins the RESTATEMENT -- any later edit of oracle/ that changes a number fails tests/test_golden_models.py, and the
torch restatement and the HIP path are checked against the same stored numbers.
    python tests/golden/make_models_golden.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import np_ref  # noqa: E402
import model_cases  # noqa: E402

out = {}
for name in model_cases.CASES:
    out[name] = {k: {"shape": list(np.shape(v)), "values": [float(t) for t in np.asarray(v, dtype=np.float64).ravel()]}
                 for k, v in model_cases.case_outputs(name, np_ref).items()}
json.dump(out, open(os.path.join(HERE, "models_kat.json"), "w"), indent=0, sort_keys=True)
print("wrote", {k: list(v) for k, v in out.items()})
