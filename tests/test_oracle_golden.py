"""The oracle's fusion module against golden vectors captured from the reference class itself
(tests/golden/make_fuse_golden.py; reference model/my_slowfast.py:260-344).  Bit-exact in fp32."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle.my_slowfast import RefFusionBuilder

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "fuse_fast_to_slow_*.npz")))


def _load_module(z):
    mod = RefFusionBuilder(8).create_module(int(z["fusion_dim_in"]), 0)
    sd = {k[len("state/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state/")}
    assert set(sd) == set(mod.state_dict().keys())       # same key scheme incl. the dead branches
    mod.load_state_dict(sd, strict=True)
    return mod


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_fuse_matches_reference_fixture(path):
    assert GOLD, "golden fixtures missing"
    z = np.load(path)
    mod = _load_module(z)
    x_s = torch.from_numpy(z["x_slow"]).requires_grad_(True)
    x_f = torch.from_numpy(z["x_fast"]).requires_grad_(True)
    mod.eval()
    with torch.no_grad():
        out_eval = mod([x_s, x_f])[0]
    assert torch.equal(out_eval, torch.from_numpy(z["out_eval"]))
    mod.train()
    out = mod([x_s, x_f])
    assert out[1] is x_f and bool(z["out_fast_is_input"])
    assert torch.equal(out[0].detach(), torch.from_numpy(z["out_train"]))
    (out[0] * torch.from_numpy(z["g"])).sum().backward()
    assert torch.equal(x_s.grad, torch.from_numpy(z["grad_x_slow"]))
    assert torch.equal(x_f.grad, torch.from_numpy(z["grad_x_fast"]))
    assert torch.equal(mod.conv_fast_to_slow[0].weight.grad, torch.from_numpy(z["grad_conv"]))
    assert torch.equal(mod.norm[0].weight.grad, torch.from_numpy(z["grad_bn_weight"]))
    assert torch.equal(mod.norm[0].bias.grad, torch.from_numpy(z["grad_bn_bias"]))
    assert torch.equal(mod.norm[0].running_mean, torch.from_numpy(z["run_mean_after"]))
    assert torch.equal(mod.norm[0].running_var, torch.from_numpy(z["run_var_after"]))
    dead = [p.grad is None for k, p in mod.named_parameters() if k.startswith(("residual", "res_unit"))]
    assert all(dead) and bool(z["dead_have_no_grad"])
