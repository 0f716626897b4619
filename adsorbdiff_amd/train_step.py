"""Score-matching training step of the PaiNN denoiser on the device (BASELINE config 5).

Replaces, for this model, what the reference's trainer does per step
(adsorbdiff/trainers/sde_denoising_trainer.py:370-537, 675-728; base_trainer.py:787-820):

    tr_so3_schedule (noising)  ->  model forward  ->  _compute_loss  ->  loss.backward()
    ->  clip_grad_norm_  ->  AdamW.step  ->  EMA.update           (+ DDP gradient all-reduce)

Every arithmetic operation of forward, loss, backward and optimizer runs in hand-written HIP kernels behind the C ABI
(csrc/train.hip, csrc/gemm.hip, csrc/graph.hip); this module only sequences the calls and owns the buffers (PyTorch as
allocator / stream / collective plumbing).  There is no autograd graph: `PaiNNTrainStep.loss_and_grad` writes the
gradients straight into ``param.grad`` of the mirror module, so ``torch.nn.parallel``-style code, checkpoints and the
reference's parameter naming keep working.

Status (round 4): products on the f16-rate matrix cores (f16x3 forward / data gradients, three-term bf16 split for the weight
gradients), message forward AND backward through fused kernels that regenerate the radial projection (nothing of size
[E, 3H] is kept from the forward), elementwise kernels unfused.  Pinned against the reference's own autograd
(tests/golden/train_small.npz, train_full.npz: loss + gradients, oracle/make_golden.py sections 6 and 10).
"""
from __future__ import annotations

import os

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import lib as _lib
from .so3_tables import Igso3Tables


class _Ops:
    """Thin typed wrappers over the adf_op_* entry points (device pointers in, status out)."""

    def __init__(self, device) -> None:
        self.lib = _lib.load()
        self.dev = torch.device(device)
        self._scratch = torch.empty(0, device=self.dev)

    def s(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def scratch(self, n: int) -> torch.Tensor:
        if self._scratch.numel() < n:
            self._scratch = torch.empty(int(n * 1.25) + 1024, device=self.dev)
        return self._scratch

    def new(self, *shape) -> torch.Tensor:
        return torch.empty(*shape, dtype=torch.float32, device=self.dev)

    def linear(self, A, W, b, M, N, K, lda=None, out=None, ldc=None):
        if out is None:
            out = self.new(M, N)
        _lib.check(self.lib.adf_op_linear_fwd(A.data_ptr(), lda or K, W.data_ptr(), b.data_ptr() if b is not None else None,
                                              out.data_ptr(), ldc or N, M, N, K, self.s()))
        return out

    def linear_bwd(self, A, W, dC, M, N, K, dW, db, want_dA=True, dA=None, lda=None, ldc=None, ldda=None, acc_dA=False):
        if want_dA and dA is None:
            dA = self.new(M, K)
        sc = self.scratch(int(self.lib.adf_op_linear_bwd_scratch(M, N, K)))
        _lib.check(self.lib.adf_op_linear_bwd(
            A.data_ptr() if A is not None else None, lda or K, W.data_ptr(), dC.data_ptr(), ldc or N,
            dA.data_ptr() if want_dA else None, ldda or K, 1 if acc_dA else 0,
            dW.data_ptr() if dW is not None else None, db.data_ptr() if db is not None else None, 1, M, N, K,
            sc.data_ptr(), self.s()))
        return dA


class PaiNNTrainStep:
    """loss + gradients of the score-matching objective for a mirror ``PaiNN`` module on a ROCm device."""

    def __init__(self, model, device="cuda:0", igso3: Optional[Igso3Tables] = None) -> None:
        self.model = model
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("PaiNNTrainStep needs a ROCm device (the HIP path has no CPU fallback)")
        if not model.so3_denoising:
            raise NotImplementedError("the training step is written for so3_denoising=True (two heads)")
        self.ops = _Ops(self.dev)
        self.lib = self.ops.lib
        self.igso3 = igso3 or Igso3Tables.shared()
        # forward of the message block through the sampler's fused kernel (ADF_TRAIN_MSG=plain: the rbfh-reading kernel)
        self.fused_message_forward = os.environ.get("ADF_TRAIN_MSG", "fused") != "plain"
        # backward of the message block with rbfh regenerated inside the kernel (message_bwd.hip) instead of kept from the
        # forward: needs the fused forward (it builds the layer's rbf_proj images) and the f16x3 arithmetic;
        # ADF_TRAIN_MSG_BWD=plain: the rbfh-reading kernel
        self.fused_message_backward = self.fused_message_forward and os.environ.get("ADF_TRAIN_MSG_BWD", "fused") != "plain"
        self._bwd_perm = None
        # rbf_proj's weight gradient without d(rbfh) [E, 3H] in memory (csrc/rbf_wgrad.hip: the fused backward stores
        # nothing per edge and the weight-gradient kernel forms d(rbfh) again while it stages its product);
        # ADF_TRAIN_RBF_WGRAD=materialised: the round-4 path (d(rbfh) written by the backward, read by adf_op_linear_bwd)
        self.rbf_wgrad_fused = os.environ.get("ADF_TRAIN_RBF_WGRAD", "fused") != "materialised"

    # ------------------------------------------------------------------ helpers
    def _params(self) -> Dict[str, torch.nn.Parameter]:
        return dict(self.model.named_parameters())

    # parameters the score path never reads: the reference's autograd leaves their .grad at None (DDP runs with
    # find_unused_parameters, base_trainer.py:442-447) and torch.optim.AdamW then skips them - no weight decay either
    UNUSED_PREFIXES = ("out_energy.",)

    def zero_grad(self) -> None:
        for name, p in self.model.named_parameters():
            if not p.requires_grad:
                continue
            if name.startswith(self.UNUSED_PREFIXES):
                p.grad = None
            elif p.grad is None:
                p.grad = torch.zeros_like(p)
            else:
                p.grad.zero_()

    # ------------------------------------------------------------------ the step
    def loss_and_grad(self, batch, targets: dict, grads_ready=None) -> torch.Tensor:
        """``batch``: noised batch on the device (pos, atomic_numbers, tags, batch, natoms, cell); ``targets``: tr_sigma
        [B,1], rot_sigma [B,1], tr_score [B,3], rot_score [B,3] (what tr_so3_schedule attaches to the batch).
        Accumulates into ``param.grad`` (call zero_grad first, like optimizer.zero_grad) and returns the device tensor
        (loss, translation term, rotation term).  ``grads_ready(names)`` (optional) is called as soon as the gradients of a
        group of parameters are final: both heads, then each layer from the last to the first, then the embedding."""
        m, ops, lib = self.model, self.ops, self.lib
        P = self._params()
        H, L, R = m.hidden_channels, m.num_layers, m.num_rbf
        eng = m.engine(self.dev, refresh=False)
        h = eng.handle
        prep = eng.prepare(batch)
        E = eng.build_graph(batch, prep)
        N, B = prep.num_atoms, prep.num_systems
        if prep.tags is None:
            raise ValueError("batch.tags is required (tag 2 marks the adsorbate)")
        s = ops.s
        G = {k: p.grad for k, p in P.items() if p.requires_grad and not k.startswith(self.UNUSED_PREFIXES)}
        for k, g in G.items():
            if g is None:
                raise RuntimeError("call zero_grad() before loss_and_grad()")
        scales = m.scale_factors()
        fused_bwd = self.fused_message_backward and bool(lib.adf_op_message_bwd_fused_supported(h))
        if fused_bwd and self._bwd_perm is None:
            perm = (C.c_int32 * (3 * H))()
            _lib.check(lib.adf_op_message_bwd_perm(h, perm, 3 * H))
            self._bwd_perm = torch.tensor(list(perm), dtype=torch.long, device=self.dev)

        # ---------------- forward with saved activations
        x = ops.new(N, H)
        _lib.check(lib.adf_op_embed_fwd(h, prep.atomic_numbers.data_ptr(), N, x.data_ptr(), s()))
        eng.check_flags()
        rbf = ops.new(E, R)
        _lib.check(lib.adf_op_rbf(h, rbf.data_ptr(), s()))
        vec = None
        saved: List[dict] = []
        for l in range(L):
            mp, up = f"message_layers.{l}.", f"update_layers.{l}."
            a = {"x": x, "vec": vec}
            a["y"], a["stats"] = ops.new(N, H), ops.new(N, 2)
            _lib.check(lib.adf_op_layernorm_fwd(x.data_ptr(), P[mp + "x_layernorm.weight"].data_ptr(),
                                                P[mp + "x_layernorm.bias"].data_ptr(), a["y"].data_ptr(),
                                                a["stats"].data_ptr(), N, H, s()))
            a["h0"] = ops.linear(a["y"], P[mp + "x_proj.0.weight"], P[mp + "x_proj.0.bias"], N, H, H)
            a["c"] = ops.new(N, H)
            _lib.check(lib.adf_op_ssilu_fwd(a["h0"].data_ptr(), a["c"].data_ptr(), N * H, s()))
            a["xh"] = ops.linear(a["c"], P[mp + "x_proj.2.weight"], P[mp + "x_proj.2.bias"], N, 3 * H, H)
            if not fused_bwd:  # kept for the rbfh-reading backward (6 KB per edge and layer)
                a["rbfh"] = ops.linear(rbf, P[mp + "rbf_proj.weight"], P[mp + "rbf_proj.bias"], E, 3 * H, R)
            a["x1"], a["vec1"] = ops.new(N, H), ops.new(N, 3, H)
            if self.fused_message_forward:
                # the sampler's fused kernel (no read of the 6 KB-per-edge rbfh)
                _lib.check(lib.adf_op_message_fwd_fused(h, l, a["xh"].data_ptr(), vec.data_ptr() if vec is not None else None,
                                                        x.data_ptr(), a["x1"].data_ptr(), a["vec1"].data_ptr(),
                                                        1 if vec is None else 0, s()))
            else:
                _lib.check(lib.adf_op_message_fwd(h, a["xh"].data_ptr(), vec.data_ptr() if vec is not None else None,
                                                  a["rbfh"].data_ptr(), x.data_ptr(), a["x1"].data_ptr(), a["vec1"].data_ptr(),
                                                  1 if vec is None else 0, s()))
            a["vv"] = ops.linear(a["vec1"], P[up + "vec_proj.weight"], None, 3 * N, 2 * H, H)
            a["cat"], a["dot"] = ops.new(N, 2 * H), ops.new(N, H)
            _lib.check(lib.adf_op_copy_rows(a["x1"].data_ptr(), H, a["cat"].data_ptr(), 2 * H, N, H, 0, s()))
            _lib.check(lib.adf_op_vdot_fwd(a["vv"].data_ptr(), a["dot"].data_ptr(), a["cat"].data_ptr() + 4 * H, 2 * H, N, H,
                                           C.c_float(1e-8), s()))
            a["u0"] = ops.linear(a["cat"], P[up + "xvec_proj.0.weight"], P[up + "xvec_proj.0.bias"], N, H, 2 * H)
            a["ua"] = ops.new(N, H)
            _lib.check(lib.adf_op_ssilu_fwd(a["u0"].data_ptr(), a["ua"].data_ptr(), N * H, s()))
            a["a"] = ops.linear(a["ua"], P[up + "xvec_proj.2.weight"], P[up + "xvec_proj.2.bias"], N, 3 * H, H)
            x2, vec2 = ops.new(N, H), ops.new(N, 3, H)
            _lib.check(lib.adf_op_update_out_fwd(a["x1"].data_ptr(), a["vec1"].data_ptr(), a["a"].data_ptr(),
                                                 a["dot"].data_ptr(), a["vv"].data_ptr(), C.c_float(scales[l]),
                                                 x2.data_ptr(), vec2.data_ptr(), N, H, s()))
            saved.append(a)
            x, vec = x2, vec2

        heads = []
        outs = []
        for hname in ("out_forces", "out_forces2"):
            hs = {}
            xin, vin, Cin = x, vec, H
            for blk, Cout in ((0, H // 2), (1, 1)):
                q = f"{hname}.output_network.{blk}."
                b = {"xin": xin, "vin": vin, "Cin": Cin, "Cout": Cout}
                b["t1"] = ops.linear(vin, P[q + "vec1_proj.weight"], None, 3 * N, Cin, Cin)
                b["cat"] = ops.new(N, 2 * Cin)
                _lib.check(lib.adf_op_copy_rows(xin.data_ptr(), Cin, b["cat"].data_ptr(), 2 * Cin, N, Cin, 0, s()))
                _lib.check(lib.adf_op_vnorm_fwd(b["t1"].data_ptr(), b["cat"].data_ptr() + 4 * Cin, 2 * Cin, N, Cin, s()))
                b["t2"] = ops.linear(vin, P[q + "vec2_proj.weight"], None, 3 * N, Cout, Cin)
                b["g0"] = ops.linear(b["cat"], P[q + "update_net.0.weight"], P[q + "update_net.0.bias"], N, Cin, 2 * Cin)
                b["ga"] = ops.new(N, Cin)
                _lib.check(lib.adf_op_ssilu_fwd(b["g0"].data_ptr(), b["ga"].data_ptr(), N * Cin, s()))
                b["o"] = ops.linear(b["ga"], P[q + "update_net.2.weight"], P[q + "update_net.2.bias"], N, 2 * Cout, Cin)
                b["xs"], b["vout"] = ops.new(N, Cout), ops.new(N, 3, Cout)
                _lib.check(lib.adf_op_gate_fwd(b["o"].data_ptr(), b["t2"].data_ptr(), b["xs"].data_ptr(), Cout,
                                               b["vout"].data_ptr(), N, Cout, s()))
                hs[blk] = b
                xin, vin, Cin = b["xs"], b["vout"], Cout
            heads.append(hs)
            outs.append(vin.reshape(N, 3))
        f1, f2 = outs

        # ---------------- loss and its gradient with respect to the two heads' outputs
        t = {k: targets[k].to(self.dev, torch.float32).contiguous() for k in ("tr_sigma", "rot_sigma", "tr_score", "rot_score")}
        rot_norm = self.igso3.score_norm(t["rot_sigma"].reshape(-1).cpu()).to(self.dev).contiguous()
        loss = ops.new(3)
        df1, df2 = ops.new(N, 3), ops.new(N, 3)
        _lib.check(lib.adf_op_score_loss(f1.data_ptr(), f2.data_ptr(), prep.tags.data_ptr(), prep.atom_offset.data_ptr(),
                                         t["tr_sigma"].data_ptr(), t["rot_sigma"].data_ptr(), t["tr_score"].data_ptr(),
                                         t["rot_score"].data_ptr(), rot_norm.data_ptr(), loss.data_ptr(), df1.data_ptr(),
                                         df2.data_ptr(), B, ops.scratch(2 * B + 16).data_ptr(), s()))

        # ---------------- backward: heads
        dx = torch.zeros(N, H, device=self.dev)
        dvec = torch.zeros(N, 3, H, device=self.dev)
        for hname, hs, dout in (("out_forces", heads[0], df1), ("out_forces2", heads[1], df2)):
            dxs, dv = None, dout.reshape(N, 3, 1)
            for blk in (1, 0):
                b = hs[blk]
                q = f"{hname}.output_network.{blk}."
                Cin, Cout = b["Cin"], b["Cout"]
                d_o, dt2 = ops.new(N, 2 * Cout), ops.new(N, 3, Cout)
                _lib.check(lib.adf_op_gate_bwd(b["o"].data_ptr(), b["t2"].data_ptr(), dxs.data_ptr() if dxs is not None else None,
                                               Cout, dv.data_ptr(), d_o.data_ptr(), dt2.data_ptr(), N, Cout, s()))
                dga = ops.linear_bwd(b["ga"], P[q + "update_net.2.weight"], d_o, N, 2 * Cout, Cin,
                                     G[q + "update_net.2.weight"], G[q + "update_net.2.bias"])
                dg0 = ops.new(N, Cin)
                _lib.check(lib.adf_op_ssilu_bwd(b["g0"].data_ptr(), dga.data_ptr(), dg0.data_ptr(), N * Cin, s()))
                dcat = ops.linear_bwd(b["cat"], P[q + "update_net.0.weight"], dg0, N, Cin, 2 * Cin,
                                      G[q + "update_net.0.weight"], G[q + "update_net.0.bias"])
                dt1 = ops.new(N, 3, Cin)
                _lib.check(lib.adf_op_vnorm_bwd(b["t1"].data_ptr(), b["cat"].data_ptr() + 4 * Cin, 2 * Cin,
                                                dcat.data_ptr() + 4 * Cin, 2 * Cin, dt1.data_ptr(), N, Cin, s()))
                # gradient of this block's inputs: x from the left half of dcat, v from the two projections
                if blk == 1:
                    dxs_in, dv_in = ops.new(N, Cin), ops.new(N, 3, Cin)
                    tgt_x, ldx, tgt_v, acc = dxs_in, Cin, dv_in, False
                else:
                    tgt_x, ldx, tgt_v, acc = dx, H, dvec, True
                _lib.check(lib.adf_op_copy_rows(dcat.data_ptr(), 2 * Cin, tgt_x.data_ptr(), ldx, N, Cin, 1 if acc else 0, s()))
                ops.linear_bwd(b["vin"], P[q + "vec1_proj.weight"], dt1, 3 * N, Cin, Cin, G[q + "vec1_proj.weight"], None,
                               dA=tgt_v, acc_dA=acc)
                ops.linear_bwd(b["vin"], P[q + "vec2_proj.weight"], dt2, 3 * N, Cout, Cin, G[q + "vec2_proj.weight"], None,
                               dA=tgt_v, acc_dA=True)
                if blk == 1:
                    dxs, dv = dxs_in, dv_in

        if grads_ready is not None:
            grads_ready(["out_forces.", "out_forces2."])
        # ---------------- backward: layers, last to first
        edge_owner = rbf_image = None
        for l in range(L - 1, -1, -1):
            a = saved[l]
            mp, up = f"message_layers.{l}.", f"update_layers.{l}."
            # update block
            da, ddot, dv1 = ops.new(N, 3 * H), ops.new(N, H), ops.new(N, 3, H)
            dx1, dvec1 = ops.new(N, H), ops.new(N, 3, H)
            _lib.check(lib.adf_op_update_out_bwd(a["a"].data_ptr(), a["dot"].data_ptr(), a["vv"].data_ptr(),
                                                 C.c_float(scales[l]), dx.data_ptr(), dvec.data_ptr(), da.data_ptr(),
                                                 ddot.data_ptr(), dv1.data_ptr(), dx1.data_ptr(), dvec1.data_ptr(), N, H, s()))
            dua = ops.linear_bwd(a["ua"], P[up + "xvec_proj.2.weight"], da, N, 3 * H, H, G[up + "xvec_proj.2.weight"],
                                 G[up + "xvec_proj.2.bias"])
            du0 = ops.new(N, H)
            _lib.check(lib.adf_op_ssilu_bwd(a["u0"].data_ptr(), dua.data_ptr(), du0.data_ptr(), N * H, s()))
            dcat = ops.linear_bwd(a["cat"], P[up + "xvec_proj.0.weight"], du0, N, H, 2 * H, G[up + "xvec_proj.0.weight"],
                                  G[up + "xvec_proj.0.bias"])
            _lib.check(lib.adf_op_copy_rows(dcat.data_ptr(), 2 * H, dx1.data_ptr(), H, N, H, 1, s()))
            dvv = ops.new(N, 3, 2 * H)
            _lib.check(lib.adf_op_vdot_bwd(a["vv"].data_ptr(), a["cat"].data_ptr() + 4 * H, 2 * H, ddot.data_ptr(),
                                           dcat.data_ptr() + 4 * H, 2 * H, dv1.data_ptr(), dvv.data_ptr(), N, H, s()))
            ops.linear_bwd(a["vec1"], P[up + "vec_proj.weight"], dvv, 3 * N, 2 * H, H, G[up + "vec_proj.weight"], None,
                           dA=dvec1, acc_dA=True)
            # message block
            first = a["vec"] is None
            dxh = ops.new(N, 3 * H)
            dvec_in = None if first else ops.new(N, 3, H)
            dx_in = ops.new(N, H)
            if fused_bwd and self.rbf_wgrad_fused:
                # nothing per edge leaves the backward; the bias gradient arrives as per-atom column sums
                db_rows = ops.new(N, 3 * H)
                _lib.check(lib.adf_op_message_bwd_fused(h, l, a["xh"].data_ptr(), a["vec"].data_ptr() if not first else None,
                                                        dx1.data_ptr(), dvec1.data_ptr(), dxh.data_ptr(), None, E,
                                                        dvec_in.data_ptr() if not first else None, dx_in.data_ptr(),
                                                        1 if first else 0, db_rows.data_ptr(), s()))
                if edge_owner is None:   # once per step: the graph and the radial basis are the same for every layer
                    edge_owner = torch.empty(E, dtype=torch.int32, device=self.dev)
                    _lib.check(lib.adf_op_edge_owner(h, edge_owner.data_ptr(), E, s()))
                    rbf_image = torch.empty(int(lib.adf_op_rbf_image_bytes(E)), dtype=torch.uint8, device=self.dev)
                    _lib.check(lib.adf_op_rbf_image(h, rbf.data_ptr(), E, rbf_image.data_ptr(), s()))
                sc = ops.scratch(int(lib.adf_op_rbf_wgrad_fused_scratch(h)))
                _lib.check(lib.adf_op_rbf_wgrad_fused(h, a["xh"].data_ptr(), a["vec"].data_ptr() if not first else None,
                                                      rbf_image.data_ptr(), edge_owner.data_ptr(), E,
                                                      G[mp + "rbf_proj.weight"].data_ptr(), sc.data_ptr(), 1 if first else 0, s()))
                ops.linear_bwd(None, P[mp + "rbf_proj.weight"], db_rows, N, 3 * H, R, None, G[mp + "rbf_proj.bias"],
                               want_dA=False)
            elif fused_bwd:
                drbfh = ops.new(E + 1, 3 * H)  # (+1: the fused kernel's spare row)
                # drbfh comes back with its columns in the kernel's own order: the weight gradient is formed on that order
                # and its rows are permuted back (3H x R, tiny)
                _lib.check(lib.adf_op_message_bwd_fused(h, l, a["xh"].data_ptr(), a["vec"].data_ptr() if not first else None,
                                                        dx1.data_ptr(), dvec1.data_ptr(), dxh.data_ptr(), drbfh.data_ptr(), E,
                                                        dvec_in.data_ptr() if not first else None, dx_in.data_ptr(),
                                                        1 if first else 0, None, s()))
                dwp = torch.zeros(3 * H, R, dtype=torch.float32, device=self.dev)
                dbp = torch.zeros(3 * H, dtype=torch.float32, device=self.dev)
                ops.linear_bwd(rbf, P[mp + "rbf_proj.weight"], drbfh, E, 3 * H, R, dwp, dbp, want_dA=False)
                G[mp + "rbf_proj.weight"].index_add_(0, self._bwd_perm, dwp)
                G[mp + "rbf_proj.bias"].index_add_(0, self._bwd_perm, dbp)
            else:
                drbfh = ops.new(E, 3 * H)
                _lib.check(lib.adf_op_message_bwd(h, a["xh"].data_ptr(), a["vec"].data_ptr() if not first else None,
                                                  a["rbfh"].data_ptr(), dx1.data_ptr(), dvec1.data_ptr(), dxh.data_ptr(),
                                                  drbfh.data_ptr(), dvec_in.data_ptr() if not first else None,
                                                  dx_in.data_ptr(), 1 if first else 0, s()))
                ops.linear_bwd(rbf, P[mp + "rbf_proj.weight"], drbfh, E, 3 * H, R, G[mp + "rbf_proj.weight"],
                               G[mp + "rbf_proj.bias"], want_dA=False)
            dc = ops.linear_bwd(a["c"], P[mp + "x_proj.2.weight"], dxh, N, 3 * H, H, G[mp + "x_proj.2.weight"],
                                G[mp + "x_proj.2.bias"])
            dh0 = ops.new(N, H)
            _lib.check(lib.adf_op_ssilu_bwd(a["h0"].data_ptr(), dc.data_ptr(), dh0.data_ptr(), N * H, s()))
            dy = ops.linear_bwd(a["y"], P[mp + "x_proj.0.weight"], dh0, N, H, H, G[mp + "x_proj.0.weight"],
                                G[mp + "x_proj.0.bias"])
            dlw, dlb = ops.new(H), ops.new(H)
            _lib.check(lib.adf_op_layernorm_bwd(a["x"].data_ptr(), P[mp + "x_layernorm.weight"].data_ptr(),
                                                a["stats"].data_ptr(), dy.data_ptr(), dx_in.data_ptr(), dlw.data_ptr(),
                                                dlb.data_ptr(), N, H, ops.scratch(512 * 2 * H + 16).data_ptr(), s()))
            _lib.check(lib.adf_op_copy_rows(dlw.data_ptr(), H, G[mp + "x_layernorm.weight"].data_ptr(), H, 1, H, 1, s()))
            _lib.check(lib.adf_op_copy_rows(dlb.data_ptr(), H, G[mp + "x_layernorm.bias"].data_ptr(), H, 1, H, 1, s()))
            dx, dvec = dx_in, dvec_in
            if grads_ready is not None:
                grads_ready([mp, up])
        _lib.check(lib.adf_op_embed_bwd(dx.data_ptr(), prep.atomic_numbers.data_ptr(),
                                        G["atom_emb.embeddings.weight"].data_ptr(), N, H, s()))
        if grads_ready is not None:
            grads_ready(["atom_emb."])
        self.last_outputs = (f1, f2)
        return loss


class FusedAdamW:
    """AdamW + global-norm clipping + EMA, one HIP launch per parameter tensor (csrc/train.hip: tr_adamw_kernel).
    Same update as torch.optim.AdamW(lr, betas, eps, weight_decay) preceded by clip_grad_norm_(max_norm) and followed
    by ExponentialMovingAverage.update (base_trainer.py:787-820); parameters named by ``no_decay`` get weight_decay 0
    (base_trainer.py:571-600, model.no_weight_decay())."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm: Optional[float] = None,
                 ema=None) -> None:
        self.lib = _lib.load()
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.ema = ema
        self.step_count = 0
        no_decay = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()
        self.entries = []
        for name, p in model.named_parameters():
            if p.requires_grad:
                self.entries.append((name, p, torch.zeros_like(p), torch.zeros_like(p), name in no_decay))
        self.sqnorm = None

    def step(self) -> torch.Tensor:
        """Applies the update from ``param.grad``; returns the (pre-clip) global gradient norm as a device tensor."""
        self.step_count += 1
        dev = self.entries[0][1].device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if self.sqnorm is None:
            self.sqnorm = torch.zeros(1, device=dev)
        self.sqnorm.zero_()
        for _, p, _, _, _ in self.entries:
            if p.grad is not None:
                _lib.check(self.lib.adf_op_sqnorm_accumulate(p.grad.data_ptr(), p.numel(), self.sqnorm.data_ptr(), stream))
        shadows = self.ema.shadow_params if self.ema is not None else None
        decay = 0.0
        if self.ema is not None:
            decay = self.ema.decay
            if self.ema.num_updates is not None:
                self.ema.num_updates += 1
                decay = min(decay, (1 + self.ema.num_updates) / (10 + self.ema.num_updates))
        i = 0
        for _, p, m_, v_, nodecay in self.entries:
            if p.grad is None:
                i += 1
                continue
            _lib.check(self.lib.adf_op_adamw_step(
                p.data_ptr(), p.grad.data_ptr(), m_.data_ptr(), v_.data_ptr(),
                shadows[i].data_ptr() if shadows is not None else None, p.numel(), self.sqnorm.data_ptr(),
                C.c_float(self.max_grad_norm or 0.0), C.c_float(self.lr), C.c_float(self.betas[0]), C.c_float(self.betas[1]),
                C.c_float(self.eps), C.c_float(0.0 if nodecay else self.weight_decay), self.step_count, C.c_float(decay),
                stream))
            i += 1
        return self.sqnorm.sqrt()


class GradientReducer:
    """Bucketed gradient all-reduce ISSUED FROM the backward pass (DDP over RCCL / xGMI, the reference's
    ``DistributedDataParallel(find_unused_parameters=True)``, base_trainer.py:442-447): ``ready(names)`` is called by
    ``PaiNNTrainStep.loss_and_grad`` as soon as the gradients of a group of parameters are final - the two heads first, then
    layer L-1 ... 0, the embedding last - and starts that bucket's all-reduce asynchronously (backend nccl = RCCL: the
    collective runs on the process group's own stream behind the kernels already enqueued, the remaining backward kernels
    keep the compute stream busy); ``finish()`` waits for the buckets, divides by the world size and writes the averages back.
    Parameters without a gradient on any rank (out_energy.*) are skipped consistently.  Buckets are one group each unless
    ``bucket_mb`` is smaller than a group (then the group is cut): a layer of the H = 512 model is 14.7 MB of gradients."""

    def __init__(self, model, world_size: int, bucket_mb: float = 64.0):
        self.world = int(world_size)
        self.limit = max(1, int(bucket_mb * 2**20 / 4))
        self.named = {k: p for k, p in model.named_parameters() if p.requires_grad}
        self.pending = []
        self.done_names = set()

    def _launch(self, grads) -> None:
        import torch.distributed as dist

        flat = torch.cat([g.reshape(-1) for g in grads])
        if dist.get_backend() == "gloo" and flat.is_cuda:  # test configuration: several ranks on one GPU
            host = flat.cpu()
            work = dist.all_reduce(host, op=dist.ReduceOp.SUM, async_op=True)
            self.pending.append((work, host, flat, grads))
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            self.pending.append((work, flat, flat, grads))

    def ready(self, names) -> None:
        """The gradients of the parameters ``names`` (exact names or prefixes) are final."""
        if self.world <= 1:
            return
        sel = [k for k in self.named if k not in self.done_names and any(k == n or k.startswith(n) for n in names)]
        self.done_names.update(sel)
        bucket, size = [], 0
        for k in sel:
            g = self.named[k].grad
            if g is None:
                continue
            bucket.append(g)
            size += g.numel()
            if size >= self.limit:
                self._launch(bucket)
                bucket, size = [], 0
        if bucket:
            self._launch(bucket)

    def finish(self) -> None:
        if self.world <= 1:
            return
        self.ready([""])   # whatever was not announced (a caller without hooks): everything that is left
        for work, red, flat, grads in self.pending:
            work.wait()
            if red is not flat:
                flat.copy_(red)
            flat.div_(self.world)
            o = 0
            for g in grads:
                g.copy_(flat[o : o + g.numel()].view_as(g))
                o += g.numel()
        self.pending = []
        self.done_names = set()


def allreduce_gradients(model, world_size: int, bucket_mb: float = 64.0) -> None:
    """DDP step after a finished backward: average ``param.grad`` over the ranks in flat buckets (RCCL all-reduce over xGMI
    under backend nccl).  The training step itself overlaps the buckets with the backward (``GradientReducer``)."""
    if world_size <= 1:
        return
    GradientReducer(model, world_size, bucket_mb).finish()
