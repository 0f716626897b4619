import ctypes as C, time, os, sys
import torch
from adsorbdiff_amd import lib as L
lib = L.load()
DEV = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def run(A, W, b, M, N, K, act, mode, rep):
    out = torch.empty(M, N, device=DEV)
    L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else None, out.data_ptr(), M, N, K, act, mode, rep, st))
    return out
torch.manual_seed(0)
# parity: mode 3 against mode 2 (bit-identical) and against f64
for (M, N, K) in [(9000, 640, 448 + 64), (1000, 1024, 1792), (777, 768, 1536), (5000, 128, 128), (4097, 384, 256), (300, 160, 64)]:
    A = torch.randn(M, K, device=DEV); A[::7] *= 1e-3; A[5] = 0
    W = torch.randn(N, K, device=DEV) * 0.05
    b = torch.randn(N, device=DEV)
    for act in (0, 2):
        o2 = run(A, W, b, M, N, K, act, 2, 1)
        o3 = run(A, W, b, M, N, K, act, 3, 1)
        ref = A.double() @ W.double().T + b.double()
        if act == 2: ref = torch.nn.functional.silu(ref)
        err = float(((o3.double() - ref).norm(dim=1) / ref.norm(dim=1).clamp(min=1e-30)).max())
        print(M, N, K, act, "identical to mode 2:", bool(torch.equal(o2, o3)), "max row err vs f64 %.2e" % err, flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "parity": sys.exit(0)
for (M, N, K) in [(256000, 1024, 1792), (512000, 768, 1536), (512000, 640, 1280)]:
    A = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) * 0.05
    b = torch.randn(N, device=DEV)
    for mode in (2, 3, 2, 3):
        run(A, W, b, M, N, K, 0, mode, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(A, W, b, M, N, K, 0, mode, 1); torch.cuda.synchronize(); t1 = time.perf_counter()
        run(A, W, b, M, N, K, 0, mode, 11); torch.cuda.synchronize(); t2 = time.perf_counter()
        ms = ((t2 - t1) - (t1 - t0)) / 10 * 1e3
        print("M %d N %d K %d mode %d: %.3f ms  (%.0f TF issued)" % (M, N, K, mode, ms, 6.0 * M * N * K / ms / 1e9), flush=True)
    del A
