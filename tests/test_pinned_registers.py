"""Two kernels keep values in registers they name themselves: the band-resident RoI forward issues
LDS reads into v[64:95] in one asm statement and waits for them in the next, the patch-owner RoI
backward keeps its sums in v[64 ...] for the whole launch and reaches them through the VGPR index
mode.  The tied asm operands only pin those registers at the statements' boundaries; between two
statements the compiler could legally copy them elsewhere or borrow them.  This test compiles
both files to assembly and fails if any instruction the compiler placed between the first and the
last of those statements touches the pinned range (a compiler update that starts doing so would
otherwise show up as wrong sums or a slowdown on the GPU box only)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "objectdetection_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-S", "--cuda-device-only"]


def _assembly(tmp_path, source):
    out = tmp_path / (source + ".s")
    subprocess.run([HIPCC, *FLAGS, "-o", str(out), os.path.join(CSRC, source)], check=True,
                   capture_output=True)
    return out.read_text().split("\n")


def _touches(code, lo, hi):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", code):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", code):
        regs.add(int(a))
    return any(lo <= r <= hi for r in regs)


def _scan(lines, kernel, marker, pinned):
    """{mangled name: (asm blocks with `marker`, compiler lines that touch the pinned range between
    the first and the last of them)}; pinned(name) -> (lo, hi)."""
    found = {}
    cur, inasm, block, between, seen = None, False, [], [], False
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1) if kernel in m.group(1) else None
            inasm, block, between, seen = False, [], [], False
            if cur:
                found[cur] = [0, []]
            continue
        if cur is None:
            continue
        if ".end_amdhsa_kernel" in ln or ln.startswith(".Lfunc_end"):
            cur = None
            continue
        if ";;#ASMSTART" in ln:
            inasm, block = True, []
            continue
        if ";;#ASMEND" in ln:
            inasm = False
            if any(marker in b for b in block):
                found[cur][0] += 1
                found[cur][1] += between          # lines since the previous marked block
                between, seen = [], True
            continue
        if inasm:
            block.append(ln)
            continue
        code = ln.split(";")[0].strip()
        lo, hi = pinned(cur)
        if seen and code and _touches(code, lo, hi):
            between.append(code)
    return found


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_band_forward_staging_registers_are_left_alone(tmp_path):
    found = _scan(_assembly(tmp_path, "frcnn_roi_band.hip"), "roi_fwd_band_kernel", "ds_read_b128",
                  lambda name: (64, 95))
    assert len(found) >= 3, found.keys()
    for name, (nblocks, bad) in found.items():
        assert nblocks > 0, name
        assert not bad, (name, bad[:5])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_patch_backward_sums_are_left_alone(tmp_path):
    def pinned(name):
        m = re.search(r"roi_bwd_patch_kernelILi(\d+)ELi(\d+)E", name)
        cells = int(m.group(1)) * int(m.group(2))
        return 64, 64 + 4 * cells + 3              # the sums and the spare group behind them

    lines = _assembly(tmp_path, "frcnn_roi_bwd.hip")
    # the index mode writes M0 without telling the compiler (hipcc does not honour an m0 clobber):
    # nothing it generates in this file may read or write M0
    assert not [ln for ln in lines if re.search(r"\bm0\b", ln.split(";")[0])]
    found = _scan(lines, "roi_bwd_patch_kernel", "s_set_gpr_idx_on", pinned)
    assert len(found) >= 2, found.keys()
    for name, (nblocks, bad) in found.items():
        assert nblocks >= 32, (name, nblocks)
        assert not bad, (name, bad[:5])
