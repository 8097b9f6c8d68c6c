"""Operator GEMM (libgcdm_ops.so: gops::k_gemm, fp32 MFMA) on the shapes of a 64-molecule QM9 training step, beside torch.matmul (rocBLAS / hipBLASLt) for orientation."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("bio-diffusion_amd")
ops = pkg.ops
dev = torch.device("cuda")
E, N = 23104, 1216
shapes = [("msg0 scalar_out", E, 605, 256), ("msg1-3 scalar_out", E, 273, 256), ("gate", E, 256, 32), ("vector_down", E * 3, 80, 20), ("vector_up", E * 3, 8, 32),
          ("ff Linear 1", N, 537, 256), ("ff Linear 2", N, 256, 256), ("attention", E, 256, 1)]
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
for name, M, K, Nn in shapes:
    x = torch.randn(M, K, device=dev, requires_grad=True); w = torch.randn(Nn, K, device=dev, requires_grad=True); b = torch.randn(Nn, device=dev, requires_grad=True)
    dy = torch.randn(M, Nn, device=dev)
    fwd = t(lambda: ops.linear(x, w, b))
    def fb():
        y = ops.linear(x, w, b); y.backward(dy)
        x.grad = w.grad = b.grad = None
    both = t(fb)
    tf = t(lambda: torch.nn.functional.linear(x, w, b))
    def tfb():
        y = torch.nn.functional.linear(x, w, b); y.backward(dy)
        x.grad = w.grad = b.grad = None
    tboth = t(tfb)
    gf = 2.0 * M * K * Nn / 1e9
    print(f"{name:20s} [{M:6d} x {K:4d}] -> {Nn:4d}: {gf:7.3f} GFLOP   ops fwd {fwd:7.1f} us ({gf / fwd * 1e3:6.1f} TF)  fwd+bwd {both:7.1f} us ({3 * gf / both * 1e3:6.1f} TF)   torch fwd {tf:7.1f} us  fwd+bwd {tboth:7.1f} us", flush=True)
