import sys, time, ctypes as C, torch
sys.path.insert(0, ".")
from adsorbdiff_amd import lib as L
lib = L.load()
dev = "cuda:0"
torch.manual_seed(0)
def run(M, N, K, mode, rep=20):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    Cm = torch.empty(M, N, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cm.data_ptr(), M, N, K, 0, mode, 1, st))
    ref = (A.double() @ W.double().T + b.double())
    err = float((Cm.double() - ref).norm() / ref.norm())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cm.data_ptr(), M, N, K, 0, mode, rep + 1, st))
    t1 = time.perf_counter()
    L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cm.data_ptr(), M, N, K, 0, mode, 1, st))
    t2 = time.perf_counter()
    dt = ((t1 - t0) - (t2 - t1)) / rep
    print(f"M={M} N={N} K={K} mode={mode}: {dt*1e3:.3f} ms  {2*M*N*K/dt/1e12:.1f} TF/s f32-eq  ({6*M*N*K/dt/1e12:.0f} f16 issued)  rel err {err:.2e}")
for (M, N, K) in ((262144, 1024, 1792), (524288, 768, 1536), (524288, 640, 1280), (262144, 896, 448), (524288, 1536, 384)):
    for mode in (1, 2):
        run(M, N, K, mode)
