"""Static census of the compiled gfx950 kernels (no GPU needed): per kernel, instruction counts by issue class plus the resource lines of
the kernel descriptor.  A proxy for where a VALU-bound kernel spends its issue slots; loop bodies are counted once (static count).

    python tools/isa_census.py [kernel-name-substring ...]  >  profiles/rNN_isa_census.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bio-diffusion_amd", "csrc", "gcdm_api.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "--cuda-device-only", "-S"]

CLASSES = [
    ("mfma", re.compile(r"^v_mfma")),
    ("trans", re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")),
    ("valu_cvt_mix", re.compile(r"^v_(cvt|fma_mix)")),
    ("valu_other", re.compile(r"^v_")),
    ("lds", re.compile(r"^ds_")),
    ("vmem_load", re.compile(r"^(global|buffer|flat)_load")),
    ("vmem_store_atomic", re.compile(r"^(global|buffer|flat)_(store|atomic)")),
    ("smem", re.compile(r"^s_(load|buffer_load)")),
    ("waitcnt", re.compile(r"^s_waitcnt")),
    ("barrier", re.compile(r"^s_barrier")),
    ("branch", re.compile(r"^s_(cbranch|branch)")),
    ("salu_other", re.compile(r"^s_")),
]


def main():
    want = sys.argv[1:] or ["k_edge_msg_x3", "k_node_x3"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        if not any(w in dem for w in want):
            continue
        cnt = collections.Counter()
        for line in body.split("\n"):
            ins = line.strip().split(" ")[0].split("\t")[0]
            if not ins or ins.startswith((";", ".", "_")) or ins.endswith(":"):
                continue
            for cls, rx in CLASSES:
                if rx.match(ins):
                    cnt[cls] += 1
                    break
        res = {k: re.search(rf"\.amdhsa_{k}\s+(\S+)", body) for k in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size", "accum_offset")}
        print(dem)
        print("   " + "  ".join(f"{k}={v.group(1)}" for k, v in res.items() if v))
        tot = sum(cnt.values())
        print("   " + "  ".join(f"{c}={cnt[c]}" for c, _ in CLASSES if cnt[c]) + f"  total={tot}")
        valu = cnt["trans"] + cnt["valu_cvt_mix"] + cnt["valu_other"]
        print(f"   static VALU : MFMA ratio = {valu / max(cnt['mfma'], 1):.1f}\n")


if __name__ == "__main__":
    main()
