"""Launch-gap accounting from a rocprofv3 --kernel-trace CSV (is a HIP-graph / persistent step worth it?).

    python tools/trace_gaps.py gpurun_out/<dir>/*/*_kernel_trace.csv [first_step last_step]

Over a window of 15 consecutive sampler steps in the timed loop: wall span, sum of kernel
durations, sum of the idle gaps between consecutive kernels of the stream, kernels per sampler step.
"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# window = a run of consecutive sampler steps in the middle of the timed loop: from the k_sample that ends step `lo` to the one that ends step `hi`
ks = [i for i, (_, _, n) in enumerate(rows) if n.startswith("k_sample")]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(ks) // 3
hi = int(sys.argv[3]) if len(sys.argv) > 3 else lo + 15
rows = rows[ks[lo] + 1:ks[hi] + 1]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
steps = sum(1 for _, _, n in rows if n.startswith("k_sample"))
print(f"kernels {len(rows)}  sampler steps {steps}  kernels/step {len(rows) / max(steps, 1):.1f}")
print(f"span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms ({100 * busy / span:.1f} %)  idle gaps {sum(gaps) / 1e6:.3f} ms ({100 * sum(gaps) / span:.1f} %)")
print(f"per step: span {span / max(steps, 1) / 1e3:.1f} us  gaps {sum(gaps) / max(steps, 1) / 1e3:.1f} us  mean gap {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us  max gap {max(gaps) / 1e3:.1f} us")
