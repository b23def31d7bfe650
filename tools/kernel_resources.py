#!/usr/bin/env python3
"""Register / LDS / spill figures of every kernel in one csrc/*.hip file, from the metadata hipcc writes into the .s
(usage: tools/kernel_resources.py conv_wgrad.hip [name-substring]).  The .s is left in /tmp for reading (s_waitcnt placement etc.)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
path = src if os.path.exists(src) else os.path.join(ROOT, "video-classification_amd", "csrc", src)
out = os.path.join("/tmp", os.path.basename(path).replace(".hip", ".s"))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                "-I" + os.path.dirname(path), "-S", "--cuda-device-only", path, "-o", out] + os.environ.get("SFK_EXTRA_FLAGS", "").split(),
               check=True)
text = open(out).read()
for blk in re.findall(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", text, flags=re.S):
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if pat not in name:
        continue
    get = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
    print(f"{dem[:78]:78s} vgpr {get('vgpr_count'):>4s} agpr {get('agpr_count'):>3s} spill {get('vgpr_spill_count'):>3s} "
          f"scratch {get('private_segment_fixed_size'):>4s} lds {get('group_segment_fixed_size'):>7s} sgpr {get('sgpr_count'):>3s}")
print(out)
