# usage: bash tools/resource_usage.sh mofa_video_amd/csrc/igemm8.hip   -> kernel, VGPRs, spills, scratch per kernel (gfx950)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage "$1" ${@:2} -o /dev/null 2>&1 |
  python3 -c '
import re, sys
name = None
row = {}
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        if name: print(name[:70], row)
        name, row = m.group(1), {}
    for k in ("VGPRs", "AGPRs", "SGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", l)
        if m: row[k.split(" [")[0]] = int(m.group(1))
if name: print(name[:70], row)
'
