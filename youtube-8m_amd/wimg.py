"""Resident operand images of the weight matrices (csrc/wimg.hip, csrc/optim.hip adam_tile_kernel; VERDICT r4 #3a).

The reference reads a variable in ``tf.matmul`` and writes it in ``AdamOptimizer.apply_gradients`` (W/train.py:459-466).  On
this chip the matrix pipe reads *operand images* (csrc/gemm_x3.hip) and rounds 2-4 re-made the images of every weight matrix
every step, once per orientation.  ``WeightImages`` gives the parameter arena of a ``Graph`` the images its model actually asks
for and lets the optimiser pass rewrite them where it rewrites the weight:

* the library notes every split of arena memory (``yt8m_wimg_watch``): after a step the owner knows which (variable, row window,
  orientation, planes, scale) images the model uses -- no per-model declarations;
* at the next ``Graph.begin_step`` those images are allocated, registered with the library's lookup table (from then on
  ``yt8m_gemm_auto_grouped`` / ``yt8m_lstm_stack_*`` / ``ops.x3_split`` / ``ops.bf16_image`` find them instead of splitting) and
  built once (``yt8m_adam_tiles(do_adam=0)``);
* ``ops.sqnorm_and_adam`` updates the owning tensors with ``yt8m_adam_tiles`` (bitwise the chunk kernel's arithmetic + the images
  in the same pass) and everything else with the chunk kernel (``yt8m_adam_multi_ex`` skips the flagged tensors);
* any torch-side write to the arena (checkpoint restore, a test injecting weights, ``dist.broadcast``) bumps the arena tensor's
  version counter; ``begin_step`` sees it and refreshes every image before the step's first product;
* a write through a RAW pointer (``yt8m_comm_broadcast_f32`` on ``data_ptr`` -- ``parallel.CabiComm.broadcast`` -- or any C-ABI host
  writing the arena) bumps nothing: the writer calls ``WeightImages.refresh()`` (rebuild now, in stream order behind the write)
  or ``invalidate()`` (rebuild at the next ``begin_step``).  ``parallel.GradReducer.attach`` does so after either broadcast
  (ADVICE r5).

``YT8M_WIMG=0`` turns the whole mechanism off (every product splits its weight operand per step, as in rounds 2-4).
"""
import bisect
import ctypes
import os

import torch

from . import _lib

ENABLED = os.environ.get("YT8M_WIMG", "1") != "0"
MIN_ELEMS = int(os.environ.get("YT8M_WIMG_MIN_ELEMS", 1 << 16))   # smaller matrices rarely reach the image kernels
# Round 6: resident HALF-PLANE ("h2", planes = 2) images of weights -- two IEEE-half planes under a power-of-two scale derived from a word
# that sits 256 bytes IN FRONT of the image: max |w| as the previous optimiser pass measured it (the split aims at 2^14 of the half
# range's 2^16: this step's weights fit unless they quadrupled).  The word's second half collects the next maximum.  Demands come from the
# products that declared the h2 role for a weight operand (yt8m_gemm_auto_grouped with YT8M_GEMM_ROLE_H2: the MoE logits).
H2 = os.environ.get("YT8M_WIMG_H2", "1") != "0"
H2_HEADER = 256
BUFFERS = {}                   # image device pointer -> (uint8 tensor, rows, K, planes): what ops.* wrap as an X3Image
STATS = {"builds": 0, "refreshes": 0, "tile_launches": 0}


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class WeightImages(object):
    def __init__(self, graph):
        self.g = graph
        self.lo = graph.params.data_ptr()
        self.hi = self.lo + max(graph.total, 1) * 4
        self.seen = 0                       # demand generation of THIS arena already examined (yt8m_wimg_demand_generation)
        self.version = None                 # arena version counter the images were last made for
        self.keys = {}                      # (tensor, row0, rows, trans, planes, scale) -> image tensor
        self.jobs_host = None
        self.jobs_dev = None
        self.job_tensor = []                # tensor index of job i (ascending)
        self.tile_base = [0]
        self.skip_dev = None
        self.rejected = set()
        self.h2_whole = {}                  # key -> the [header | image] tensor of an h2 image (keys[k] is the view behind the header)
        self.watching = False
        if ENABLED and graph.params.is_cuda and graph.total > 0:
            _lib.check(_lib.lib().yt8m_wimg_watch(ctypes.c_void_p(self.lo), ctypes.c_void_p(self.hi), 1))
            self.watching = True

    # ---- what the step asked for ---------------------------------------------------------------------------------------------
    def _demands(self):
        L = _lib.lib()
        n = L.yt8m_wimg_demands(None, 0)
        arr = (_lib.WimgDemand * max(n, 1))()
        n = min(n, L.yt8m_wimg_demands(arr, n))
        return [arr[i] for i in range(n) if self.lo <= (arr[i].src or 0) < self.hi]

    def _variable_at(self, addr):
        tv = self.g.trainable_variables()
        off = (addr - self.lo) // 4
        i = bisect.bisect_right(self._offsets, off) - 1
        if i < 0:
            return None
        v = tv[i]
        return v if v.offset <= off < v.offset + v.numel() else None

    def _key_of(self, d):
        """(tensor, row0, rows, trans, planes, scale) of a demand the tile kernel can serve, else None."""
        v = self._variable_at(d.src)
        if v is None or v.data.dim() != 2 or v.numel() < MIN_ELEMS:
            return None
        R, C = v.data.shape
        rel = (d.src - self.lo) // 4 - v.offset
        if d.C != C or d.ld != C or rel % C != 0:
            return None
        row0, rows = rel // C, d.R
        if row0 % 64 != 0 or row0 + rows > R or not (rows % 64 == 0 or row0 + rows == R):
            return None
        if d.planes not in (1, 3) and not (H2 and d.planes == 2 and d.scale == 0.0):
            return None
        return (v.index, int(row0), int(rows), int(d.trans), int(d.planes), float(d.scale))

    # ---- begin_step ------------------------------------------------------------------------------------------------------------
    def begin_step(self):
        """Called by Graph.begin_step on a finalized graph: turns newly seen demands into resident images, refreshes the images
        after a torch-side write to the arena."""
        if not self.watching:
            return
        if self._generation() != self.seen:
            self._extend()
        if self.jobs_dev is not None and self.g.params._version != self.version:
            self.refresh()

    def _generation(self):
        return _lib.lib().yt8m_wimg_demand_generation(ctypes.c_void_p(self.lo), ctypes.c_void_p(self.hi))

    def invalidate(self):
        """The arena was written behind torch's back (raw-pointer broadcast, a C-ABI host): every image is rebuilt at the next
        begin_step."""
        self.version = None

    def _extend(self):
        g = self.g
        L = _lib.lib()
        self._offsets = [v.offset for v in g.trainable_variables()]
        self.seen = self._generation()
        fresh = []
        for d in self._demands():
            k = self._key_of(d)
            if k is None or k in self.keys or k in self.rejected:
                continue
            fresh.append(k)
        if not fresh:
            return
        self.add(fresh)

    def add(self, keys):
        """Allocates, builds and registers the images `keys` ((tensor, row0, rows, trans, planes, scale) each; at most four distinct
        (row window, planes, scale) per tensor -- the rest is left to the per-step split)."""
        g = self.g
        L = _lib.lib()
        tv = g.trainable_variables()
        specs = {}
        for k in self.keys:
            specs.setdefault(k[0], set()).add((k[1], k[2], k[4], k[5]))
        fresh = []
        for k in keys:
            have = specs.setdefault(k[0], set())
            sp = (k[1], k[2], k[4], k[5])
            if k in self.keys:                                             # already resident: nothing to do (and nothing to reject)
                continue
            if sp not in have and len(have) >= 4:
                self.rejected.add(k)
                continue
            have.add(sp)
            fresh.append(k)
        if not fresh:
            return
        for k in fresh:
            t, row0, rows, trans, planes, scale = k
            C = tv[t].data.shape[1]
            img_rows, K = (C, rows) if trans else (rows, C)
            nbytes = L.yt8m_x3_image_bytes(img_rows, K) // 3 * planes
            if planes == 2:                                                # [256-byte header: scale word, next maximum | image]
                whole = torch.zeros(H2_HEADER + max(nbytes, 16), dtype=torch.uint8, device=g.params.device)
                buf = whole[H2_HEADER:]
                self.h2_whole[k] = whole
            else:
                buf = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=g.params.device)
            self.keys[k] = buf
            BUFFERS[buf.data_ptr()] = (buf, img_rows, K, planes)
        self._layout()
        for k in fresh:                                                    # visible to the consumers only once they are built
            t, row0, rows, trans, planes, scale = k
            v = tv[t]
            C = v.data.shape[1]
            src = ctypes.c_void_p(self.lo + 4 * (v.offset + row0 * C))
            _lib.check(L.yt8m_wimg_register(src, rows, C, C, trans, planes, scale, _p(self.keys[k])))
        STATS["builds"] += 1

    def _layout(self):
        g = self.g
        L = _lib.lib()
        tv = g.trainable_variables()
        per_tensor = {}
        for k in self.keys:
            per_tensor.setdefault(k[0], {}).setdefault((k[1], k[2], k[4], k[5]), {})[k[3]] = self.keys[k]
        tensors = sorted(per_tensor)
        jobs = (_lib.WimgJob * len(tensors))()
        for j, t in enumerate(tensors):
            v = tv[t]
            R, C = v.data.shape
            jobs[j].offset, jobs[j].R, jobs[j].C, jobs[j].tensor = v.offset, R, C, t
            specs = sorted(per_tensor[t])
            jobs[j].nspec = len(specs)
            for i, (row0, rows, planes, scale) in enumerate(specs):
                imgs = per_tensor[t][(row0, rows, planes, scale)]
                sp = jobs[j].spec[i]
                sp.plain = imgs[0].data_ptr() if 0 in imgs else None
                sp.trans = imgs[1].data_ptr() if 1 in imgs else None
                sp.row0, sp.rows, sp.scale, sp.planes = row0, rows, (1.0 if planes == 2 else scale), planes   # (h2: scale 0 = "device word")
        total = L.yt8m_wimg_jobs_layout(jobs, len(tensors))
        if total < 0:
            _lib.check(int(total))
        self._previous = (self.jobs_dev, self.skip_dev)                    # a launch still in flight may read the old tables
        self.jobs_host = jobs
        self.job_tensor = tensors
        self.tile_base = [jobs[j].tile_base for j in range(len(tensors))] + [int(total)]
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8)
        self.jobs_dev = raw.to(g.params.device)
        skip = torch.zeros(max(len(tv), 1), dtype=torch.uint8)
        skip[tensors] = 1
        self.skip_dev = skip.to(g.params.device)
        self.refresh()

    # ---- the two passes ----------------------------------------------------------------------------------------------------------
    def _tiles(self, j0, j1, do_adam, hyper, stream):
        g = self.g
        if j1 <= j0:
            return
        jobs = ctypes.c_void_p(self.jobs_dev.data_ptr() + ctypes.sizeof(_lib.WimgJob) * j0)
        gscale, clip, lr_t, b1, b2, eps = hyper
        _lib.check(_lib.lib().yt8m_adam_tiles(_p(g.params), _p(g.adam_m), _p(g.adam_v), _p(g.grads), jobs, j1 - j0, self.tile_base[j0],
                                              self.tile_base[j1] - self.tile_base[j0], _p(g.l2), gscale, _p(g.norms), clip, lr_t, b1, b2,
                                              eps, int(do_adam), stream))
        STATS["tile_launches"] += 1

    def refresh(self):
        """Rebuilds every image from the weights as they are (first build; after a torch-side write to the arena)."""
        if self.jobs_dev is None:
            return
        from .ops import _stream
        if self.h2_whole:                                                  # h2 images: the maximum of the weights as they are -> the header's
            L = _lib.lib()                                                 # "next" word; the tile pass's roll makes it the scale word
            tv = self.g.trainable_variables()
            for k, whole in self.h2_whole.items():
                t, row0, rows, trans, planes, scale = k
                v = tv[t]
                C = v.data.shape[1]
                whole[:H2_HEADER].zero_()
                src = ctypes.c_void_p(self.lo + 4 * (v.offset + row0 * C))
                _lib.check(L.yt8m_h2_absmax(src, rows, C, C, ctypes.c_void_p(whole.data_ptr() + 4), _stream()))
        self._tiles(0, len(self.job_tensor), False, (1.0, 0.0, 0.0, 0.0, 0.0, 0.0), _stream())
        self.version = self.g.params._version
        STATS["refreshes"] += 1

    def adam(self, lo, hi, hyper, stream):
        """clip + Adam + images of the image-owning tensors in [lo, hi) (the chunk pass skipped them: skip_dev)."""
        j0 = bisect.bisect_left(self.job_tensor, lo)
        j1 = bisect.bisect_left(self.job_tensor, hi)
        self._tiles(j0, j1, True, hyper, stream)

    @property
    def active(self):
        return self.jobs_dev is not None

    # ---- release -----------------------------------------------------------------------------------------------------------------
    def close(self):
        if not self.watching:
            return
        self.watching = False
        try:
            L = _lib.lib()
            L.yt8m_wimg_unregister(ctypes.c_void_p(self.lo), ctypes.c_void_p(self.hi))
            L.yt8m_wimg_watch(ctypes.c_void_p(self.lo), ctypes.c_void_p(self.hi), 0)
        except Exception:                                                  # interpreter shutdown
            pass
        for buf in self.keys.values():
            BUFFERS.pop(buf.data_ptr(), None)
        self.keys = {}
        self.h2_whole = {}
        self.jobs_dev = self.skip_dev = None


def resident_image(x, R, C, ld, trans, planes, scale=1.0):
    """(uint8 buffer, rows, K) of the resident image of fp32 x[R, C] (row pitch ld) in the asked orientation, or None."""
    if not BUFFERS:
        return None
    ptr = _lib.lib().yt8m_wimg_lookup(ctypes.c_void_p(x.data_ptr()), R, C, ld, int(trans), int(planes), float(scale))
    if not ptr:
        return None
    ent = BUFFERS.get(ptr)
    return None if ent is None else ent[:3]
