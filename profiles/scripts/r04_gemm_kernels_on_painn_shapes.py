# PaiNN node-product shapes through the three f16x3 product kernels of the library (run under `rocprofv3 --kernel-trace`, then
# read the kernel durations): gemm16.hip (PaiNN, fp32 A split while staged), eqv2_gemm16.hip 256x256 fp32-A and pre-split kernels.
import sys, ctypes as C, torch
sys.path.insert(0, ".")
from adsorbdiff_amd import lib as L
lib = L.load()
dev = "cuda:0"
torch.manual_seed(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K) in ((200000, 512, 512), (200000, 1536, 512), (200000, 512, 1024), (600000, 1024, 512)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    Cm = torch.empty(M, N, device=dev)
    for _ in range(3):
        L.check(lib.adf_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cm.data_ptr(), M, N, K, 0, 1, st))
        for mode in (1, 2):
            L.check(lib.adf_eqv2_linear_forward(A.data_ptr(), W.data_ptr(), b.data_ptr(), Cm.data_ptr(), M, N, K, 0, mode, 1, st))
    torch.cuda.synchronize()
