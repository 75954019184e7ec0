"""ORACLE (test infrastructure only -- never imported by the product path).

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123 library's
philox4x32_R(10, ctr, key)) restated in numpy integer arithmetic, plus the two random elementwise ops the device library
derives from it (youtube-8m_amd/csrc/random.hip):

  dropout  -- tf.nn.dropout(x, keep_prob): x / keep_prob * floor(keep_prob + u)
              (W/all_video_models/deep_combine_chain_model.py:57-58; DropoutWrapper(input_keep_prob) of
              W/all_frame_models/lstm_memory_model.py:36-45)
  noise    -- x + N(0, stddev^2) (W/all_frame_models/lstm_memory_model.py:62-63)

Pinning: the block function is checked against the three known-answer vectors published with Random123
(tests/test_oracle_thirdparty.py).  The mapping "element e -> word (e & 3) of block (e >> 2)" and u = (word >> 8) * 2^-24 are
this build's own convention (TF-1.0 draws from different streams; the reference pins no random values)."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Counters: uint64-typed numpy arrays holding 32-bit words; keys: Python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3)]
    for _ in range(10):
        p0 = c0 * np.uint64(M0)
        p1 = c2 * np.uint64(M1)
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def words(n, seed, offset=0):
    """The uint32 word of each of the n elements starting at logical element `offset`."""
    e = np.arange(offset, offset + n, dtype=np.uint64)
    g = e >> np.uint64(2)
    zero = np.zeros_like(g)
    r = philox4x32_10(g & np.uint64(MASK), g >> np.uint64(32), zero, zero, seed & MASK, (seed >> 32) & MASK)
    sel = (e & np.uint64(3)).astype(np.int64)
    return np.choose(sel, r)


def uniform01(n, seed, offset=0):
    return (words(n, seed, offset) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def dropout_mask(n, keep_prob, seed, offset=0):
    return (np.float32(keep_prob) + uniform01(n, seed, offset)) >= np.float32(1.0)


def dropout(x, keep_prob, seed, offset=0):
    x = np.asarray(x, dtype=np.float32)
    m = dropout_mask(x.size, keep_prob, seed, offset).reshape(x.shape)
    return np.where(m, x / np.float32(keep_prob), np.float32(0.0)).astype(np.float32)


def normal(n, seed, offset=0):
    """Box-Muller over word pairs of one block: words (0,1) -> elements 0,1; words (2,3) -> elements 2,3 (float64 maths)."""
    e = np.arange(offset, offset + n, dtype=np.uint64)
    base = (e & ~np.uint64(1)).astype(np.int64)
    lo = int(base.min()) if n else 0
    span = int(base.max()) + 2 - lo if n else 0
    u = uniform01(span, seed, lo).astype(np.float64)
    ia = base - lo
    ra = np.sqrt(-2.0 * np.log(1.0 - u[ia]))
    th = 2.0 * np.pi * u[ia + 1]
    return np.where((e & np.uint64(1)) == 0, ra * np.cos(th), ra * np.sin(th))


def add_noise(x, stddev, seed, offset=0):
    x = np.asarray(x, dtype=np.float64)
    return x + stddev * normal(x.size, seed, offset).reshape(x.shape)
