"""Per-key gradient error of the depth-26 mini model (identity blocks) vs the oracle, on the GPU, for several input seeds:
separates a wiring error (O(1), many keys, every seed) from ReLU/arg-max flip noise (one small key, seed dependent)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from helpers import rel_l2
from oracle import my_slowfast as o
from test_engine_cpu import engine_grads_as_state_dict, make_models, oracle_train_step_with_engine_mask
from video_classification_amd._lib import HipBackend

for seed in (5, 6, 7):
    for ref_style in (True, False):
        om, m = make_models(ref_style, device="cuda", backend=HipBackend(), depth=26)
        g = torch.Generator().manual_seed(seed)
        if ref_style:
            x = o.prepare_slowfast_data(torch.randn(2, 4, 21, 64, 64, generator=g))
        else:
            x = o.pack_pathway(torch.randn(2, 3, 8, 64, 64, generator=g))
        m.train()
        labels = torch.tensor([1, 4])
        y_o, _ = oracle_train_step_with_engine_mask(om, m.engine, x, labels)
        y_m = m([t.cuda() for t in x])
        torch.nn.functional.cross_entropy(y_m, labels.cuda()).backward()
        gsd = engine_grads_as_state_dict(m.engine)
        errs = sorted(((rel_l2(gsd[k].cpu(), p.grad), k, p.grad.numel()) for k, p in om.named_parameters() if p.grad is not None),
                      reverse=True)
        print(f"seed {seed} ref_style {ref_style}: fwd err {float((y_m.detach().cpu() - y_o).abs().max()):.2e}; worst keys:")
        for e, k, n in errs[:4]:
            print(f"    {e:.3e}  {k} ({n})")
