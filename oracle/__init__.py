"""TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the wangheda/youtube-8m hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package.  The product (``youtube-8m_amd/``) never imports it and has no CPU
fallback: it fails loudly when the HIP library is missing.

Pinning status
--------------
* ``oracle.metrics`` (GAP@20 / Hit@1 / PERR / AP): PINNED against known answers produced by the
  *imported* reference metric code (``tests/golden/make_golden.py`` -> ``tests/golden/metrics_kat.json``).
* Everything else (``np_ref`` / ``torch_ref``: dequantise, L2-normalise, logistic, MoE, chained MoE,
  BasicLSTM / dynamic_rnn, attention pooling, DBoF, cross-entropy, clip, TF-Adam, LR decay):
  **parity unpinned** -- the reference is Python-2 / TensorFlow-1.0 graph code that cannot be imported
  or run here (no python2, no TF wheel, no network) and the reference ships no tests or golden
  vectors.  The arithmetic is restated from the cited reference lines plus TF-1.0 documented
  semantics; the two restatements (numpy fp64, torch autograd) are written
  independently and must agree with each other.  ``tests/test_oracle_thirdparty.py`` additionally checks them against
  PyTorch's own implementations of the same published algorithms (torch.nn.LSTM on packed sequences, torch.optim.Adam as
  eps -> 0, torch.nn.functional): that pins "standard algorithm", it does NOT pin TensorFlow's behaviour.
* NetVLAD is not in the reference at all; its oracle is this
  package's own definition.
"""
