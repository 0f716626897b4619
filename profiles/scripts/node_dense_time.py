"""Time of the PaiNN node products (update block + the message block's two products) at `nsys` systems, per layer pass:
HIP-event categories of the engine (node_dense, message) over `reps` repetitions of message_layer + update_layer."""
import os, sys
sys.path.insert(0, '.')
import torch
from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS
from adsorbdiff_amd.synthetic import make_batch
nsys = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = "cuda:0"
torch.manual_seed(0)
m = PaiNN(None, 50, 1, cutoff=10.0, max_neighbors=50, scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True).to(dev).eval()
eng = m.engine()
b = make_batch(nsys, seed=1000).to(dev)
eng.build_graph(b)
H = m.hidden_channels
x = m.atom_emb.embeddings.weight.detach()[b.atomic_numbers.long() - 1].contiguous()
vec = torch.randn(x.shape[0], 3, H, device=dev) * 0.01
for _ in range(2):
    x1, v1 = eng.message_layer(1, x, vec); eng.update_layer(1, x1, v1)
torch.cuda.synchronize()
eng.profile_enable(True)
for _ in range(6):
    x1, v1 = eng.message_layer(1, x, vec); eng.update_layer(1, x1, v1)
torch.cuda.synchronize()
pr = eng.profile_read()
print("stagger", os.environ.get("ADF_GEMM_STAGGER", "auto"), os.environ.get("ADF_GEMM_STAGGER_GAIN", ""), "node_dense ms/layer", round(pr["node_dense"][0] / 6, 3), "message ms/layer", round(pr["message"][0] / 6, 3))
