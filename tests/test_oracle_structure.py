"""Structural pins of the oracle (last row): parameter counts, conv MACs, key scheme."""
import torch

from oracle import my_slowfast as o


def _nparams(m):
    return sum(p.numel() for p in m.parameters())


def test_param_counts():
    assert _nparams(o.canonical_slowfast_8x8(400)) == 34_566_488
    assert _nparams(o.canonical_slowfast_8x8(249)) == 34_218_433
    ref = o.init_my_slowfast(249, (5, 15), (64, 8))
    assert _nparams(ref) == 38_077_321
    dead = sum(p.numel() for k, p in ref.named_parameters() if ".residual." in k or ".res_unit." in k)
    assert dead == 4_025_160
    per_stage = [sum(p.numel() for k, p in ref.named_parameters()
                     if k.startswith(f"blocks.{i}.multipathway_fusion.") and (".residual." in k or ".res_unit." in k))
                 for i in range(4)]
    assert per_stage == [12_200, 191_840, 765_120, 3_056_000]


def test_state_dict_keys_of_checkpoint_surgery():
    # the 12 keys train.py:94-108 deletes from the Kinetics checkpoint must exist under these names
    ref = o.init_my_slowfast(249, (5, 15), (64, 8))
    keys = set(ref.state_dict().keys())
    must = ["blocks.0.multipathway_blocks.0.conv.weight", "blocks.0.multipathway_blocks.1.conv.weight",
            "blocks.6.proj.weight", "blocks.6.proj.bias"]
    for b in (1, 2, 3, 4):
        must += [f"blocks.{b}.multipathway_blocks.0.res_blocks.0.branch1_conv.weight",
                 f"blocks.{b}.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight"]
    assert all(k in keys for k in must)
    sd = ref.state_dict()
    assert tuple(sd["blocks.0.multipathway_blocks.0.conv.weight"].shape) == (64, 5, 1, 7, 7)
    assert tuple(sd["blocks.0.multipathway_blocks.1.conv.weight"].shape) == (8, 15, 1, 7, 7)
    assert tuple(sd["blocks.1.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight"].shape) == (64, 80, 1, 1, 1)
    assert tuple(sd["blocks.3.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight"].shape) == (256, 640, 3, 1, 1)
    assert tuple(sd["blocks.6.proj.weight"].shape) == (249, 2304)
    # reference fusion keys carry the ModuleList index, canonical ones do not
    assert "blocks.0.multipathway_fusion.conv_fast_to_slow.0.weight" in keys
    can = set(o.canonical_slowfast_8x8(400).state_dict().keys())
    assert "blocks.0.multipathway_fusion.conv_fast_to_slow.weight" in can
    assert tuple(o.canonical_slowfast_8x8(400).state_dict()[
        "blocks.0.multipathway_fusion.conv_fast_to_slow.weight"].shape) == (16, 8, 7, 1, 1)
    assert not any("multipathway_fusion" in k and k.startswith("blocks.4") for k in keys)


def test_conv_macs_canonical():
    m = o.canonical_slowfast_8x8(400)
    x = o.pack_pathway(torch.zeros(1, 3, 32, 224, 224))
    assert [tuple(t.shape) for t in x] == [(1, 3, 8, 224, 224), (1, 3, 32, 224, 224)]
    macs = o.conv_macs(m, x)
    assert abs(macs / 1e9 - 50.309) < 2e-3


def test_pack_pathway_indices():
    f = torch.arange(32.0).view(1, 1, 32, 1, 1)
    s, fast = o.pack_pathway(f)
    assert s.flatten().tolist() == [0, 4, 8, 13, 17, 22, 26, 31]
    assert fast is f


def test_ref_forward_shapes_small():
    # REF wiring at reduced size: same T on both pathways, head pools (4,2,2) stride 1, multi-position head
    m = o.init_my_slowfast(249, (5, 15), (64, 8)).eval()
    clips = torch.randn(1, 8, 21, 64, 64)
    x = o.prepare_slowfast_data(clips)
    assert x[0].shape == (1, 5, 8, 64, 64) and x[1].shape == (1, 15, 8, 64, 64)
    feats = []
    h = m.blocks[5].register_forward_hook(lambda mod, i, out: feats.append(out.shape))
    with torch.no_grad():
        y = m(x)
    h.remove()
    assert y.shape == (1, 249)
    assert tuple(feats[0]) == (1, 2304, 5, 1, 1)
