"""The hot kernels must not touch scratch memory: hipcc spills silently (hoisted division reciprocals, pointer-phi allocas,
accumulator tuples merged at a branch -- DESIGN.md section 3 lists the cases met), and a spill inside a K loop or an epilogue
costs more than most optimisations gain.  Cross-compiles the four hot sources for gfx950 with
-Rpass-analysis=kernel-resource-usage (no GPU needed) and checks every shipped kernel instantiation; the diagnostic K-loop
variants of tools/igemm8_probe.py (template parameter VAR != 0) are exempt."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
SOURCES = ["attention.hip", "igemm.hip", "igemm8.hip", "igemm320.hip"]


def _usage(src):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "mofa_video_amd", "csrc", src), "-o", os.devnull]
    out = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, name = {}, None
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            rows[name] = {}
        m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", line)
        if m and name:
            rows[name][m.group(1).split(" [")[0]] = int(m.group(2))
    return rows


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_hot_kernels_use_no_scratch():
    with ThreadPoolExecutor(len(SOURCES)) as ex:
        results = list(ex.map(_usage, SOURCES))
    bad, seen = [], 0
    for src, rows in zip(SOURCES, results):
        assert rows, f"no kernel resource remarks for {src}"
        for name, r in rows.items():
            probe = re.search(r"igemm8_f16_kernelILi\d+ELi([1-9]\d*)E", name)     # VAR != 0: diagnostic build variants
            if probe:
                continue
            seen += 1
            if r.get("ScratchSize", 0) or r.get("VGPRs Spill", 0):
                bad.append((src, name, r))
    assert seen >= 37
    assert not bad, bad
