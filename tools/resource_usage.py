"""Prints VGPR / AGPR / scratch / occupancy per kernel from a `hipcc -Rpass-analysis=kernel-resource-usage` log (build container).

    python tools/resource_usage.py build/ab/u3.log [substring]
"""
import re
import subprocess
import sys

log = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "k_"
cur = None
rows = {}
for line in log.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w][\w \[\]/]*?): (\S+)", line)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
    if pat in name:
        print(f"{name:40s} VGPR {v.get('VGPRs')} AGPR {v.get('AGPRs')} scratch {v.get('ScratchSize [bytes/lane]')} occ {v.get('Occupancy [waves/SIMD]')} SGPR {v.get('TotalSGPRs')} LDS {v.get('LDS Size [bytes/block]')}")
