"""message3 (new) vs message (v1) kernel: outputs and time, on the benchmark batch."""
import os, sys, time
sys.path.insert(0, '.')
import torch
from adsorbdiff_amd.painn_denoising import PaiNN
from adsorbdiff_amd.scaling import PAINN_NB6_SCALE_FACTORS
from adsorbdiff_amd.synthetic import make_batch
nsys = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
def mk(kernel):
    os.environ["ADF_MSG_KERNEL"] = kernel   # read at set_weights (adf_pack_rbf)
    torch.manual_seed(0)
    m = PaiNN(None, 50, 1, cutoff=10.0, max_neighbors=50, scale_file=PAINN_NB6_SCALE_FACTORS, so3_denoising=True).to(dev).eval()
    eng = m.engine()
    return m, eng
b = make_batch(nsys, seed=1000).to(dev)
res = {}
for kern in ("v1", "v3"):
    m, eng = mk(kern)
    eng.build_graph(b)
    H = m.hidden_channels
    x = m.atom_emb.embeddings.weight.detach()[b.atomic_numbers.long() - 1].contiguous()
    vec = torch.zeros(x.shape[0], 3, H, device=dev)
    outs = []
    for li in range(2):
        x, vec = eng.message_layer(li, x.contiguous(), vec.contiguous())
        outs.append((x.clone(), vec.clone()))
        x, vec = eng.update_layer(li, x.contiguous(), vec.contiguous())
    # time layer 1's message block (vec != 0)
    xin, vin = x.contiguous(), vec.contiguous()
    for _ in range(2): eng.message_layer(1, xin, vin)
    torch.cuda.synchronize()
    eng.profile_enable(True)
    for _ in range(5): eng.message_layer(1, xin, vin)
    torch.cuda.synchronize()
    pr = eng.profile_read()
    print(kern, "profile", {k: (round(v[0] / max(v[1], 1), 3), v[1]) if isinstance(v, tuple) else v for k, v in pr.items()})
    eng.profile_enable(False)
    f1, f2 = m(b)
    res[kern] = (outs, f1.clone(), f2.clone())
for li in range(2):
    for nm, a, c in (("x", res["v1"][0][li][0], res["v3"][0][li][0]), ("vec", res["v1"][0][li][1], res["v3"][0][li][1])):
        d = (a - c).abs()
        rows_same = (d.reshape(d.shape[0], -1).max(dim=1).values == 0).float().mean().item()
        print(f"layer {li} {nm}: max abs diff {d.max().item():.3e} (ref max {a.abs().max().item():.3e}), rel {((a-c).norm()/a.norm()).item():.3e}, bit-identical rows {rows_same:.4f}, finite {bool(torch.isfinite(c).all())}")
for nm, a, c in (("f1", res["v1"][1], res["v3"][1]), ("f2", res["v1"][2], res["v3"][2])):
    print(nm, "rel", ((a - c).norm() / a.norm()).item())
