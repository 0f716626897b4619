import sys, time; sys.path.insert(0, '.')
import torch
from adsorbdiff_amd.denoising_torch import Denoiser, DiffTorchCalc
from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS
from adsorbdiff_amd.synthetic import make_batch
from adsorbdiff_amd.trainer import DenoisingTrainer
dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = PaiNN(None, 50, 1, hidden_channels=512, num_layers=6, num_rbf=128, cutoff=10.0, max_neighbors=50, scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True).eval()
tr = DenoisingTrainer(m, device=dev)
for B in (1, 4, 16, 64):
    batch = make_batch(B, seed=1000).to(dev)
    for inc in (True, False):
        params = dict(num_steps=100, ads_std_low=0.1, ads_std_high=10, rot_std_low=0.01, rot_std_high=1.55, ode=True, early_stop=False, incremental_layers=inc)
        def run():
            torch.manual_seed(0)
            den = Denoiser(batch.clone(), DiffTorchCalc(tr), params, device=str(dev))
            den.run(); torch.cuda.synchronize()
        run()
        t0 = time.perf_counter(); run(); run(); dt = (time.perf_counter() - t0) / 2
        print(f"B={B} incremental={inc}: {100 / dt:.1f} it/s  ({dt * 10:.2f} ms/step)")
