"""What does bf16 storage cost a depth-50 SlowFast train step, and does the engine pay exactly that?
One canonical 8x8 model, N clips of 3x32x224^2, train mode, the engine's dropout mask injected into the oracle.  Prints, for
two parameter states ('randomized': every BatchNorm gamma in [0.5, 1.5], the tests' randomize(); 'mild': the reference init with
the block-final gammas in [0.1, 0.3] -- a residual network that does not amplify), the per-tensor gradient cosines against the
fp32 oracle of: the engine in fp32, the engine in bf16, and the ORACLE ITSELF with every conv / BatchNorm / ReLU / pool output
and gradient rounded to bf16 (what any correct bf16 implementation looks like).
usage: python tools/probe/bf16_parity.py [N]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import my_slowfast as o                                    # noqa: E402
from test_engine_cpu import engine_grads_as_state_dict, oracle_train_step_with_engine_mask, randomize   # noqa: E402
from video_classification_amd import arch                              # noqa: E402
from video_classification_amd.slowfast import SlowFast, pack_pathway_index   # noqa: E402
from video_classification_amd.train import TrainStep                   # noqa: E402

DEV = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2


class RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).float()

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).float()


def emulate_bf16_storage(model):
    hs = []
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv3d, torch.nn.BatchNorm3d, torch.nn.ReLU, torch.nn.MaxPool3d)):
            hs.append(m.register_forward_hook(lambda mod, inp, out: RoundBF16.apply(out)))
    return hs


def mild(om, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in om.state_dict().items():
            if k.endswith("running_var"):
                v.copy_(torch.rand(v.shape, generator=g) + 0.5)
            elif k.endswith("running_mean"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.2)
            elif "norm_c.weight" in k:
                v.copy_(torch.rand(v.shape, generator=g) * 0.2 + 0.1)
            elif ".norm" in k and k.endswith("weight"):
                v.copy_(torch.rand(v.shape, generator=g) * 0.5 + 0.75)
            elif ".norm" in k and k.endswith("bias"):
                v.copy_(torch.randn(v.shape, generator=g) * 0.1)


def cosines(gsd, om):
    rows = []
    for k, p in om.named_parameters():
        a, b = gsd[k].cpu().flatten().double(), p.grad.flatten().double()
        rows.append((float(a @ b / (a.norm() * b.norm() + 1e-30)), float(a.norm() / (b.norm() + 1e-30)), k, a.numel()))
    rows.sort()
    c = np.array([r[0] for r in rows])
    return f"median {np.median(c):.4f} p10 {np.percentile(c, 10):.4f} min {c.min():.4f} ({rows[0][2]}) norm ratio {min(r[1] for r in rows):.2f}..{max(r[1] for r in rows):.2f}"


for regime in ("randomized", "mild"):
    torch.manual_seed(0)
    om = o.canonical_slowfast_8x8(400)
    (randomize if regime == "randomized" else mild)(om, 3)
    with torch.no_grad():
        for k, v in om.state_dict().items():
            if v.dim() == 5:
                v.copy_(v.to(torch.bfloat16).float())
    frames = torch.randn(N, 3, 32, 224, 224, generator=torch.Generator().manual_seed(21)).to(torch.bfloat16)
    labels = torch.randint(0, 400, (N,), generator=torch.Generator().manual_seed(5))
    sd0 = {k: v.clone() for k, v in om.state_dict().items()}
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        m = SlowFast(arch.canonical_spec(400), dtype=dtype, device=DEV)
        m.load_state_dict(sd0, strict=True)
        m.train()
        eng = m.engine
        om.load_state_dict(sd0)
        y_o, loss_o = oracle_train_step_with_engine_mask(om, eng, o.pack_pathway(frames.float()), labels)
        ref = {k: p.grad.clone() for k, p in om.named_parameters()}
        step = TrainStep(eng, lr=0.0, use_graph=False)
        fd = frames.to(DEV).to(dtype)
        idx = pack_pathway_index(32, 4, DEV)
        loss_m = float(step(fd, fd, labels.to(DEV), slow_t_index=idx))
        torch.cuda.synchronize()
        y_m = eng._plan_for(fd, fd, idx, True).logits.float().cpu()
        fwd = float((y_m - y_o).abs().max() / y_o.abs().max())
        osd = om.state_dict()
        rv = max(float((L.rv.cpu() - osd[L.cb.norm_key + ".running_var"]).abs().max() / osd[L.cb.norm_key + ".running_var"].abs().max())
                 for L in eng.layers)
        print(f"[{regime} N={N}] engine {str(dtype)[6:]:9s}: logits {fwd:.2e} loss {loss_m:.4f} vs {float(loss_o):.4f} running var {rv:.2e}; "
              f"grad cos {cosines(engine_grads_as_state_dict(eng), om)}", flush=True)
        if dtype == torch.bfloat16:
            # the oracle with bf16 storage emulated, against the plain fp32 oracle (same dropout mask: same engine seed)
            eng.drop_seed.sub_(1)
            om.load_state_dict(sd0)
            hs = emulate_bf16_storage(om)
            y_e, loss_e = oracle_train_step_with_engine_mask(om, eng, o.pack_pathway(frames.float()), labels)
            for h in hs:
                h.remove()
            emu = {k: p.grad.clone() for k, p in om.named_parameters()}
            for k, p in om.named_parameters():
                p.grad = ref[k]
            print(f"[{regime} N={N}] oracle with bf16 storage: logits {float((y_e - y_o).abs().max() / y_o.abs().max()):.2e} loss {float(loss_e):.4f}; "
                  f"grad cos {cosines(emu, om)}", flush=True)
            # engine bf16 against the emulated-bf16 oracle
            for k, p in om.named_parameters():
                p.grad = emu[k]
            print(f"[{regime} N={N}] engine bf16 vs oracle-with-bf16-storage: grad cos {cosines(engine_grads_as_state_dict(eng), om)}", flush=True)
        del m, eng, step
        torch.cuda.empty_cache()
