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


_ASM_CACHE = {}


def compile_to_asm():
    key = tuple(os.path.getmtime(os.path.join(os.path.dirname(SRC), f)) for f in sorted(os.listdir(os.path.dirname(SRC))))
    if key not in _ASM_CACHE:
        _ASM_CACHE.clear()
        _ASM_CACHE[key] = _compile_to_asm()
    return _ASM_CACHE[key]


def _compile_to_asm():
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


def _wait_states(ins):
    """Issue slots an instruction keeps the wave busy for hazard purposes: s_nop N = N + 1 states, anything else 1."""
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def _kernels(text):
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M):
        ins = []
        for line in m.group(2).split("\n"):
            t = line.strip()
            if not t or t.startswith((";", ".")) or t.split()[0].endswith(":"):
                continue
            ins.append(t)
        yield m.group(1), ins


def scratch_users(text=None, pattern="k_edge_msg_x3"):
    """Kernels (mangled names containing `pattern`) whose ISA touches scratch memory.  Round 5: ONE spilled VGPR of k_edge_msg_x3<64,16,64> -- a lane id the
    compiler had hoisted out of the persistent loop -- cost 2.4 % of every tile: its scratch reload is the youngest load in flight and, loads returning in
    order, waits for every prefetched row of the next tile.  The persistent kernels must stay spill-free."""
    text = compile_to_asm() if text is None else text
    return [name for name, ins in _kernels(text) if pattern in name and any(i.startswith("scratch_") or "offen" in i and i.startswith("buffer_") and "s[0:3]" in i for i in ins)]


def lint_partial_writes(text=None, min_states=1, embed_rule=True):
    """Measured on gfx950 (tools/mfma_partial_write_hazard.hip): an MFMA issued with NO wait state behind a 16-bit partial write
    (v_fma_mixlo/hi_f16) of one of its source registers reads the old register; one wait state is enough.  The compiler inserts that
    wait state between an inline-asm definition and its consumer, and x3_settle (gcdm_edge_x3.hip.h) adds two more wherever a split feeds
    an MFMA directly.  Checked per kernel, for every MFMA and every source register (A, B, C operands), in straight-line order:
      (1) if the register's LAST writer is a v_fma_mix*, at least `min_states` wait states (instructions, s_nop N = N + 1) lie between
          that write and the MFMA;
      (2) (`embed_rule`) in k_edge_embed_x3 -- the kernel whose stale-B-operand fault was never pinned to one instruction pair --
          no v_fma_mix* is scheduled between two MFMAs: its splits must stay fenced off the matrix pipe's bursts (the x3_settle barrier).
    Returns [(kernel, writer, mfma)] violations."""
    text = text if text is not None else compile_to_asm()
    bad = []
    for name, ins in _kernels(text):
        last = {}                      # vgpr -> (index of last writer, is a partial write)
        states = [0]                   # prefix sums of wait states
        for t in ins:
            states.append(states[-1] + _wait_states(t))
        first_mfma = last_mfma = None
        for i, t in enumerate(ins):
            op = t.split()[0]
            if op.startswith("v_mfma"):
                ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
                for r in set().union(*[_vregs(o) for o in ops[1:4]]):
                    w = last.get(r)
                    if w is not None and w[1] and states[i] - states[w[0] + 1] < min_states:
                        bad.append((name, ins[w[0]], t))
                        break
                first_mfma = i if first_mfma is None else first_mfma
                last_mfma = i
            if op.startswith(("v_", "ds_read", "buffer_load", "global_load", "flat_load")) and " " in t:
                dst = t.split(None, 1)[1].split(",")[0].strip()
                for r in _vregs(dst):
                    last[r] = (i, op.startswith("v_fma_mix"))
        if embed_rule and "k_edge_embed_x3" in name and first_mfma is not None:
            seen_mfma = False
            for i in range(first_mfma, last_mfma + 1):
                op = ins[i].split()[0]
                if op.startswith("v_mfma"):
                    seen_mfma = True
                elif op.startswith("v_fma_mix") and seen_mfma:
                    nxt = next((j for j in range(i + 1, last_mfma + 1) if ins[j].startswith("v_mfma")), None)
                    prv = next((j for j in range(i - 1, first_mfma - 1, -1) if ins[j].startswith("v_mfma")), None)
                    # inside a burst = MFMAs on both sides with no s_nop fence in between
                    if nxt is not None and prv is not None and not any(ins[j].startswith("s_nop") for j in range(prv, nxt)):
                        bad.append((name, ins[i], ins[nxt]))
                        break
    return bad


def main():
    if sys.argv[1:] == ["--lint"]:
        bad = lint_partial_writes()
        for k, w, f in bad:
            print(f"{k}:\n    {w}\n    {f}")
        print(f"{len(bad)} violation(s): MFMA issued too close behind a partial write of one of its sources, or a split inside an MFMA burst of k_edge_embed_x3")
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
