import os, sys
sys.path.insert(0, '.')
import torch
from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS
from adsorbdiff_amd.synthetic import make_batch
nsys = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
torch.manual_seed(0)
m = PaiNN(None, 50, 1, cutoff=10.0, max_neighbors=50, scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True).to(dev).eval()
eng = m.engine()
b = make_batch(nsys, seed=1000).to(dev)
eng.build_graph(b)
H = m.hidden_channels
x = m.atom_emb.embeddings.weight.detach()[b.atomic_numbers.long() - 1].contiguous()
vec = torch.randn(x.shape[0], 3, H, device=dev) * 0.01
for _ in range(2): eng.message_layer(1, x, vec)
torch.cuda.synchronize()
eng.profile_enable(True)
for _ in range(5): eng.message_layer(1, x, vec)
torch.cuda.synchronize()
pr = eng.profile_read()
print(os.environ.get("ADF_LIB_PATH", "default")[-14:], "message ms", round(pr["message"][0] / pr["message"][1], 3))
