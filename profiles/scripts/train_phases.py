import sys, time, torch
sys.path.insert(0, ".")
from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS
from adsorbdiff_amd.synthetic import make_batch
from adsorbdiff_amd.trainer import DenoisingTrainer
from adsorbdiff_amd.noising import tr_so3_schedule
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = PaiNN(None, 50, 1, hidden_channels=512, num_layers=6, num_rbf=128, cutoff=12.0, max_neighbors=50,
              scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True).to(dev)
tr = DenoisingTrainer(model, device=dev)
params = dict(ads_std_low=0.1, ads_std_high=10, free_std_low=0.0, free_std_high=0.0, rot_std_low=0.01, rot_std_high=1.55, num_steps=50)
tr.setup_training(params, lr=1e-4, weight_decay=0.0, clip_grad_norm=10.0, ema_decay=0.999)
batch = make_batch(256, seed=2000).to(dev)
for _ in range(2): tr.train_step(batch.clone())
def timeit(name, fn, n=3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print(f"{name:22s} {(time.perf_counter()-t0)/n*1e3:8.2f} ms"); return r
timeit("batch.clone", lambda: batch.clone())
nb = timeit("noising", lambda: tr_so3_schedule(batch.clone(), params, tr.train_engine.igso3))
targets = {k: getattr(nb, k) for k in ("tr_sigma", "rot_sigma", "tr_score", "rot_score")}
timeit("zero_grad", lambda: tr.train_engine.zero_grad())
timeit("loss_and_grad", lambda: tr.train_engine.loss_and_grad(nb, targets))
timeit("optimizer.step", lambda: tr.optimizer.step())
timeit("train_step", lambda: tr.train_step(batch.clone()))
