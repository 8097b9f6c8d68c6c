"""Static census of the compiled gfx950 kernels (no GPU needed): per kernel, instruction counts by issue class plus the resource lines of
the kernel descriptor.  A proxy for where a VALU-bound kernel spends its issue slots; loop bodies are counted once (static count).

    python tools/isa_census.py [kernel-name-substring ...]  >  profiles/rNN_isa_census.txt
    python tools/isa_census.py --lint                       # partial-write -> MFMA adjacency check (see lint_partial_writes)
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


def compile_to_asm():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def _vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def lint_partial_writes(text=None):
    """Measured on gfx950 (tools/mfma_partial_write_hazard.hip): an MFMA issued with NO wait state behind a 16-bit partial write
    (v_fma_mixlo/hi_f16) of one of its source registers reads the old register; one wait state is enough.  The compiler inserts that
    wait state between an inline-asm definition and its consumer -- this check fails if a build ever comes out without it.
    Returns [(kernel, writer, mfma)] violations."""
    text = text if text is not None else compile_to_asm()
    bad = []
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        ins = []
        for line in m.group(2).split("\n"):
            t = line.strip()
            if not t or t.startswith((";", ".")) or t.split()[0].endswith(":"):
                continue
            ins.append(t)
        for i in range(1, len(ins)):
            if not ins[i].startswith("v_mfma"):
                continue
            prev = ins[i - 1]
            if not prev.startswith("v_fma_mix"):
                continue
            dst = _vregs(prev.split(None, 1)[1].split(",")[0].strip())
            ops = [o.strip() for o in ins[i].split(None, 1)[1].split(",")]
            srcs = set().union(*[_vregs(o) for o in ops[1:]])
            if dst & srcs:
                bad.append((m.group(1), prev, ins[i]))
    return bad


def main():
    if sys.argv[1:] == ["--lint"]:
        bad = lint_partial_writes()
        for k, w, f in bad:
            print(f"{k}:\n    {w}\n    {f}")
        print(f"{len(bad)} MFMA(s) issued directly behind a partial write of one of their sources")
        sys.exit(1 if bad else 0)
    want = sys.argv[1:] or ["k_edge_msg_x3", "k_node_x3"]
    text = compile_to_asm()
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
