"""Static instruction census of the edge-message kernel PER PHASE: the -DGCDM_STAMPS build carries an s_memtime at every phase boundary, so the
instructions between two of them are the phase's code (both branches of uniform `if`s included: the hooked and the plain GEMM variant, the two
halves of the gate-partial fold).  Build container only (llvm-objdump of the library's gfx950 code object; no GPU):

    tools/build_variants.sh stamps:-DGCDM_STAMPS
    python tools/phase_census.py [build/ab/libgcdm_stamps.so] [kernel-symbol-substring]

DESIGN.md section 8 quotes its numbers (e.g. SiLU + gate + partial fold of a residual GCP2: ~450 instructions per wave, 64 of them transcendental)."""
import collections
import glob
import os
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
NAMES = {0: "(prologue: who am I, first tile's loads)", 1: "msg0 pre-phase", 2: "barrier", 3: "PQ add", 4: "msg0 GEMM", 5: "(stamp)", 6: "SiLU + gate + partial fold (msg0)", 7: "barrier",
         8: "state images", 9: "barrier", 10: "(stamp)", 11: "GEMM k = 1 (+ hooked vector stages; both variants)", 12: "(stamp)", 13: "SiLU + gate + partial fold", 14: "(stamp)",
         15: "residual add + state images", 16: "barrier", 17: "GCP2 k = 2, 3 (all of the above twice) + attention + fp32 image", 18: "(barrier)", 19: "segment sums"}


def main():
    lib = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "build/ab/libgcdm_stamps.so")
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_edge_msg_x3ILi64ELi16ELi64E"
    d = os.path.dirname(lib)
    subprocess.run([OBJDUMP, "-d", "--offloading", lib], capture_output=True, text=True, cwd=d)
    cos = sorted(set(glob.glob(lib + "*gfx950*")))
    out = subprocess.run([OBJDUMP, "-d", cos[0]], capture_output=True, text=True).stdout
    for c in cos:
        os.remove(c)
    lines = out.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*" + re.escape(pat) + ".*>:", l))
    body = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]+ <", l):
            break
        m = re.match(r"^\s+(\S+)", l)
        if m:
            body.append(m.group(1))
    marks = [i for i, op in enumerate(body) if op == "s_memtime"]
    print(f"{lines[start].split('<')[1][:60]}: {len(body)} instructions, {len(marks)} stamps")
    bounds = [0] + marks + [len(body)]
    for j in range(len(bounds) - 1):
        c = collections.Counter(body[bounds[j]:bounds[j + 1]])
        tot = sum(c.values())
        if tot < 40:
            continue
        trans = sum(v for k, v in c.items() if k.startswith(("v_exp", "v_rcp", "v_sqrt", "v_rsq", "v_log")))
        mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
        valu = sum(v for k, v in c.items() if k.startswith("v_")) - mfma
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        vmem = sum(v for k, v in c.items() if k.startswith(("buffer_", "global_", "scratch_")))
        top = ", ".join(f"{k}:{v}" for k, v in c.most_common(8))
        print(f"  [{j:2d}] {NAMES.get(j, ''):<72s} {tot:5d} instr  mfma {mfma:4d}  valu {valu:4d} (trans {trans:3d})  lds {lds:4d}  vmem {vmem:4d}   {top}")


if __name__ == "__main__":
    main()
