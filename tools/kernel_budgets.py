#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel in libfrcnn_hip.so, read from the code objects'
metadata notes (no GPU needed).

  python tools/kernel_budgets.py            # table
  python tools/kernel_budgets.py --json

The library embeds one gfx950 code object per source file (clang offload bundles).  They are
extracted into a temporary directory with llvm-objdump --offloading, their NT_AMDGPU_METADATA
notes printed with llvm-readelf --notes and parsed as YAML; names are demangled with c++filt.
"""
from __future__ import annotations

import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "objectdetection_amd", "lib", "libfrcnn_hip.so")


def kernel_budgets(lib_path: str = LIB):
    """{demangled kernel name: dict(vgpr, agpr, sgpr, lds, scratch, wg_max)}"""
    import yaml
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for co in sorted(glob.glob(so + ".*amdgcn*")):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True,
                                   capture_output=True, text=True).stdout
            if "amdhsa.kernels" not in notes:
                continue
            body = notes[notes.index("---"):]
            body = body[:body.index("\n...")] if "\n..." in body else body
            meta = yaml.safe_load(body)
            for k in meta.get("amdhsa.kernels", []):
                out[k[".name"]] = dict(vgpr=k[".vgpr_count"], agpr=k.get(".agpr_count", 0),
                                       sgpr=k[".sgpr_count"], lds=k[".group_segment_fixed_size"],
                                       scratch=k[".private_segment_fixed_size"],
                                       wg_max=k.get(".max_flat_workgroup_size"))
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), check=True,
                         capture_output=True, text=True).stdout.splitlines()
    res = {}
    for n, d in zip(names, dem):
        d = d.split("(")[0].replace("void ", "").replace("frcnn::", "").replace(" ", "")
        res[d] = out[n]
    return res


if __name__ == "__main__":
    b = kernel_budgets()
    if "--json" in sys.argv:
        print(json.dumps(b, indent=1, sort_keys=True))
    else:
        print(f"{'kernel':64s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'lds':>7s} {'scratch':>8s}")
        for n in sorted(b):
            v = b[n]
            print(f"{n:64s} {v['vgpr']:5d} {v['agpr']:5d} {v['sgpr']:5d} {v['lds']:7d} {v['scratch']:8d}")
