"""Not -m gpu: register allocation of the hot kernels, read from hipcc's own resource remarks (cross-compiles without a GPU).

The persistent recurrences hold their W_h slice in registers (64 VGPRs of a 168-VGPR budget at three waves per SIMD) and the
image GEMMs run at 241-247 of 256; the load / MFMA placement pinned in round 4 only pays while nothing spills --
two of the placements tried for the backward recurrence put the resident weights in scratch (32 VGPRs spilled, scratch loads in
front of every MFMA group) and would have passed every parity test."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "youtube-8m_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


def _remarks(src):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "--cuda-device-only", "-c", src,
           "-o", os.devnull, "-Rpass-analysis=kernel-resource-usage"]
    p = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


@pytest.fixture(scope="module")
def remarks():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    with ThreadPoolExecutor(2) as ex:
        a, b = ex.map(_remarks, ["lstm_persist.hip", "gemm_x3.hip"])
    return {**a, **b}


def _scratch_loads(src, names):
    """scratch_load instructions per function of `names` in the ISA of `src` (a stack object that is only ever STORED to -- the
    compiler's dead copy of a few resident weight fragments in the round-6 GRU backward kernel -- costs a handful of stores in the
    prologue and nothing in the loop; one that is loaded from is a spill by another name)."""
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "--cuda-device-only", "-S", src, "-o", "-"]
    p = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    out, cur = {}, None
    for line in p.stdout.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1) if m.group(1) in names else None
            if cur:
                out[cur] = 0
        elif cur and "scratch_load" in line:
            out[cur] += 1
        elif cur and "s_endpgm" in line:
            cur = None
    return out


def test_hot_kernels_do_not_spill(remarks):
    assert len(remarks) > 20
    # (SGPR spills go to VGPR lanes, not to memory: the image-writing backward variants have ~25 of them and no scratch)
    spilled = {k: v for k, v in remarks.items() if v.get("VGPRs Spill", 0)}
    assert not spilled, spilled
    stack = {k for k, v in remarks.items() if v.get("ScratchSize", 0)}
    assert all("gru_persist_bwd_kernel" in k for k in stack), stack            # nothing else may own a stack object at all
    if stack:
        loads = _scratch_loads("lstm_persist.hip", stack)
        assert set(loads) == stack and not any(loads.values()), loads


def test_recurrences_keep_three_waves_per_simd_and_gemms_two(remarks):
    rec = {k: v for k, v in remarks.items() if "lstm_persist_bwd_kernel" in k or "lstm_persist_fwd" in k}
    assert rec and all(v["VGPRs"] <= 168 and v["Occupancy"] >= 3 for v in rec.values()), {k: v["VGPRs"] for k, v in rec.items()}
    gemm = {k: v for k, v in remarks.items() if re.search(r"gemm_(x3q?|b1q?)_kernel", k)}
    assert len(gemm) >= 6 and all(v["VGPRs"] <= 256 and v["Occupancy"] >= 2 for v in gemm.values()), {k: v["VGPRs"] for k, v in gemm.items()}
