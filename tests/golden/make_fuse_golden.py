"""Generate tests/golden/fuse_fast_to_slow_*.npz from the REFERENCE's own fusion classes.

Run in the build container only (needs /root/reference; the reference never travels to the GPU box):

    python tests/golden/make_fuse_golden.py

The reference module /root/reference/model/my_slowfast.py imports third-party packages that are absent
here (torchvision, pytorchvideo, the dataset module -> cv2).  Only its two fusion classes
(MyFastToSlowFusionBuilder my_slowfast.py:136-257, FuseFastToSlow :260-344) are exercised, and they are pure
torch.nn, so the missing imports are replaced by empty stub modules; the single helper they use from
pytorchvideo, ``set_attributes`` (copy constructor locals onto self), is provided by the stub.

Each fixture holds: inputs x_slow/x_fast, every state-dict tensor of the module (seeded, BN running stats made
non-trivial), the train-mode output + updated running stats, the eval-mode output, and the gradients of
sum(out * g) w.r.t. inputs and live parameters.  Data only -- no reference source text is stored.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference_fusion():
    def set_attributes(self, params=None):
        if params:
            for k, v in params.items():
                if k != "self":
                    setattr(self, k, v)

    anything = lambda *a, **k: None  # noqa: E731
    _stub("torchvision")
    _stub("torchvision.transforms")
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    _stub("pytorchvideo")
    _stub("pytorchvideo.models")
    _stub("pytorchvideo.layers")
    _stub("pytorchvideo.layers.utils", set_attributes=set_attributes)
    _stub("pytorchvideo.models.slowfast", create_slowfast=anything)
    _stub("pytorchvideo.models.resnet", create_bottleneck_block=anything, create_res_stage=anything)
    _stub("pytorchvideo.models.stem", create_res_basic_stem=anything)
    _stub("pytorchvideo.models.head", create_res_basic_head=anything, create_res_roi_pooling_head=anything)
    _stub("pytorchvideo.models.net", DetectionBBoxNetwork=object, MultiPathWayWithFuse=object, Net=object)
    _stub("dataset")
    _stub("dataset.chalearn_dataset", ChalearnVideoDataset=object)
    for opt in ("tqdm", "requests", "matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image"):
        try:
            importlib.import_module(opt)
        except Exception:
            _stub(opt, tqdm=anything, Image=object)
    sys.path.insert(0, REF)
    return importlib.import_module("model.my_slowfast")


def make(ref_mod, name, fusion_dim_in, n, t, h, w, seed):
    torch.manual_seed(seed)
    builder = ref_mod.MyFastToSlowFusionBuilder.build_fusion_builder(8)
    mod = builder.create_module(fusion_dim_in, 0)
    with torch.no_grad():
        for k, v in mod.state_dict().items():
            if k.endswith("running_mean"):
                v.copy_(torch.randn_like(v) * 0.3)
            elif k.endswith("running_var"):
                v.copy_(torch.rand_like(v) + 0.5)
            elif k.endswith("num_batches_tracked"):
                pass
            elif ".norm." in k or k.startswith("norm."):
                v.copy_(torch.randn_like(v) * 0.5 + (1.0 if k.endswith("weight") else 0.0))
    state0 = {k: v.clone() for k, v in mod.state_dict().items()}
    c_fast = fusion_dim_in // 8
    x_s = torch.randn(n, fusion_dim_in, t, h, w, requires_grad=True)
    x_f = torch.randn(n, c_fast, t, h, w, requires_grad=True)

    mod.eval()
    with torch.no_grad():
        out_eval = mod([x_s, x_f])[0].clone()

    mod.train()
    out = mod([x_s, x_f])
    g = torch.randn_like(out[0])
    (out[0] * g).sum().backward()
    state1 = mod.state_dict()
    rec = {
        "fusion_dim_in": np.int64(fusion_dim_in),
        "x_slow": x_s.detach().numpy(), "x_fast": x_f.detach().numpy(), "g": g.numpy(),
        "out_eval": out_eval.numpy(), "out_train": out[0].detach().numpy(),
        "out_fast_is_input": np.bool_(out[1] is x_f),
        "grad_x_slow": x_s.grad.numpy(), "grad_x_fast": x_f.grad.numpy(),
        "grad_conv": mod.conv_fast_to_slow[0].weight.grad.numpy(),
        "grad_bn_weight": mod.norm[0].weight.grad.numpy(), "grad_bn_bias": mod.norm[0].bias.grad.numpy(),
        "run_mean_after": state1["norm.0.running_mean"].numpy(),
        "run_var_after": state1["norm.0.running_var"].numpy(),
        "dead_have_no_grad": np.bool_(all(p.grad is None for k, p in mod.named_parameters()
                                          if k.startswith("residual") or k.startswith("res_unit"))),
    }
    for k, v in state0.items():
        rec["state/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, f"fuse_fast_to_slow_{name}.npz"), **rec)
    # identity past the last stage (my_slowfast.py:181-182)
    assert isinstance(builder.create_module(2048, 4), torch.nn.Identity)
    print(name, "params", sum(p.numel() for p in mod.parameters()), "out", tuple(out[0].shape))


if __name__ == "__main__":
    ref = import_reference_fusion()
    make(ref, "c64", 64, 2, 4, 8, 8, seed=11)      # stage 0: 8 -> 16 fused channels, 64 -> 80
    make(ref, "c256", 256, 1, 3, 4, 4, seed=12)    # stage 1: 32 -> 64 fused channels, 256 -> 320
