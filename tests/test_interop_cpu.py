"""The reference's checkpoint / config / trainer interop surface EXECUTED, on the
CPU with tests/emu_backend.py standing in for libsfk:
  /root/reference/train.py:93-123   `model_state` checkpoint, 12-key surgery, strict=False load
  /root/reference/train.py:185-214  file name, lexicographic "newest", HTAH fallback, strict=True
  /root/reference/train.py:255-284  Trainer.train: eval every epoch, best-accuracy save, final save
  /root/reference/train.py:405-413  yaml merge order over config/*.yaml
  /root/reference/model/my_slowfast.py:90-92  MODEL.FUSE = False
  /root/reference/train.py:64-76    res2d plumbing (BASELINE.json config 1)
"""
import os

import numpy as np
import pytest
import torch

from emu_backend import EmuBackend
from helpers import rel_err
from oracle import my_slowfast as o
from video_classification_amd.config import get_cfg
from video_classification_amd.train import ModelManager, SyntheticChalearn, Trainer

SURGERY_KEYS = [
    'blocks.0.multipathway_blocks.0.conv.weight', 'blocks.0.multipathway_blocks.1.conv.weight',
    'blocks.6.proj.weight', 'blocks.6.proj.bias',
    'blocks.1.multipathway_blocks.0.res_blocks.0.branch1_conv.weight',
    'blocks.1.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight',
    'blocks.2.multipathway_blocks.0.res_blocks.0.branch1_conv.weight',
    'blocks.2.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight',
    'blocks.3.multipathway_blocks.0.res_blocks.0.branch1_conv.weight',
    'blocks.3.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight',
    'blocks.4.multipathway_blocks.0.res_blocks.0.branch1_conv.weight',
    'blocks.4.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight',
]   # the list of train.py:94-108, kept here as data: the product rebuilds it programmatically


def small_cfg(tmp_path, name="slowfast-Torso", depth=18, clip_len=4, crop="CropLHand", batch=2, classes=7):
    cfg = get_cfg()
    cfg.CHALEARN.ROOT = str(tmp_path)
    cfg.CHALEARN.BATCH_SIZE = batch
    cfg.CHALEARN.CLIP_LEN = clip_len
    cfg.CHALEARN.NUM_CLASS = classes
    cfg.MODEL.NAME = name
    cfg.MODEL.R3D_INPUT = crop
    cfg.MODEL.DEPTH = depth
    cfg.MODEL.LR = 2e-4
    cfg.NUM_CPU = 0
    return cfg


def small_trainer(cfg, seed=1, videos=4):
    tr = SyntheticChalearn(cfg, "train", num_videos=videos, seed=seed)
    te = SyntheticChalearn(cfg, "test", num_videos=3, clips_per_video=(1, 2), seed=seed + 1)
    return Trainer(cfg, train_set=tr, test_set=te, device="cpu", backend=EmuBackend())


# ------------------------------------------------------------------ a6: pretrained `model_state` + delete_mismatch
def test_delete_mismatch_is_the_reference_list():
    sd = {k: 0 for k in SURGERY_KEYS}
    sd["blocks.1.multipathway_blocks.1.res_blocks.0.branch2.conv_a.weight"] = 1
    out = ModelManager.delete_mismatch(sd)
    assert list(out) == ["blocks.1.multipathway_blocks.1.res_blocks.0.branch2.conv_a.weight"]
    with pytest.raises(KeyError):
        ModelManager.delete_mismatch({})              # the reference's `del` raises on a checkpoint without those keys


def test_pretrained_kinetics_checkpoint_loads_through_model_manager(tmp_path, monkeypatch):
    """pretrained/SLOWFAST_8x8_R50.pyth = {'model_state': <canonical SlowFast-R50 8x8, 400 classes>} in the cwd
    (train.py:116): 12 tensors dropped, the canonical fusion keys (`conv_fast_to_slow.weight`, `norm.*`) silently
    unmatched against the reference's ModuleList keys (`...0.weight`) under strict=False, everything else loaded."""
    torch.manual_seed(7)
    kinetics = o.canonical_slowfast_8x8(400)
    with torch.no_grad():
        for k, v in kinetics.state_dict().items():
            if k.endswith(("running_mean", "bias")):
                v.normal_(0, 0.1)                       # away from the init values (0 / 1) so "loaded" is observable
            elif k.endswith(("running_var",)) or (".norm" in k and k.endswith("weight")):
                v.uniform_(0.5, 1.5)
    state = kinetics.state_dict()
    (tmp_path / "pretrained").mkdir()
    torch.save({"model_state": state, "epoch": 196}, tmp_path / "pretrained" / "SLOWFAST_8x8_R50.pyth")
    monkeypatch.chdir(tmp_path)
    cfg = small_cfg(tmp_path, depth=50, classes=249)
    model = ModelManager(cfg, device="cpu", backend=EmuBackend()).init_model()
    got = model.state_dict()
    assert model.num_parameters() == 38_077_321
    loaded = dropped = unmatched = 0
    for k, v in state.items():
        if k in SURGERY_KEYS:
            dropped += 1
            if tuple(got[k].shape) == tuple(v.shape):   # same shape, but deleted before the load: still the init value
                assert not torch.equal(got[k], v), k
            continue
        if k not in got:
            assert ".multipathway_fusion." in k, k      # conv_fast_to_slow.weight / norm.* vs the ModuleList's .0.
            unmatched += 1
            continue
        assert torch.equal(got[k].float(), v.float()), k
        loaded += 1
    assert dropped == 12 and unmatched == 4 * 6 and loaded == len(state) - 12 - 24
    # the reference-only keys (fusion ModuleLists, dead residual / res_unit) keep their init
    ours_only = [k for k in got if k not in state]
    assert all(".multipathway_fusion." in k for k in ours_only) and len(ours_only) > 24


# ------------------------------------------------------------------ a4: save_ckpt / load_ckpt
def test_checkpoint_names_sort_order_fallback_and_strictness(tmp_path, capsys):
    cfg = small_cfg(tmp_path)
    t = small_trainer(cfg)
    ckdir = tmp_path / "logs" / "checkpoints" / "slowfast-Torso"
    P = t.model.engine.P.data
    P.fill_(0.25)
    t.save_ckpt(epoch=9, acc=0.9)
    P.fill_(0.5)
    t.save_ckpt(epoch=50, acc=0.1)
    assert sorted(os.listdir(ckdir)) == ["acc0.100_e50.ckpt", "acc0.900_e9.ckpt"]          # 'acc%.3f_e%d.ckpt'
    # the file is a flat fp32 state_dict a plain torch user can read, with the reference's tensor shapes
    sd = torch.load(ckdir / "acc0.900_e9.ckpt", weights_only=True)
    assert tuple(sd["blocks.0.multipathway_blocks.0.conv.weight"].shape) == (64, 5, 1, 7, 7)
    assert all(v.dtype in (torch.float32, torch.int64) for v in sd.values())
    # a fresh trainer resumes from the lexicographically LAST name = the best accuracy, not the latest epoch (train.py:207)
    t2 = small_trainer(cfg)
    assert "acc0.900_e9.ckpt" in capsys.readouterr().out
    assert float(t2.model.engine.P.data[0]) == 0.25
    # HTAH fallback (train.py:200-206): no checkpoint under this model's name -> slowfast-HTAH's newest one
    os.rename(ckdir, tmp_path / "logs" / "checkpoints" / "slowfast-HTAH")
    t3 = small_trainer(small_cfg(tmp_path, name="slowfast-LHand"))
    out = capsys.readouterr().out
    assert "try using HTAH" in out and "slowfast-HTAH" in out
    assert float(t3.model.engine.P.data[0]) == 0.25
    # nothing anywhere: a warning, the init weights stay
    t4 = small_trainer(small_cfg(tmp_path / "empty", name="slowfast-LHand"))
    assert "no HTAH checkpoint found" in capsys.readouterr().out
    # strict=True (train.py:212): a foreign key or a missing key is an error
    bad = dict(sd)
    bad["blocks.9.bogus.weight"] = torch.zeros(1)
    torch.save(bad, tmp_path / "logs" / "checkpoints" / "slowfast-HTAH" / "acc0.950_e1.ckpt")
    with pytest.raises(RuntimeError, match="unexpected"):
        small_trainer(small_cfg(tmp_path, name="slowfast-LHand"))
    short = dict(sd)
    del short["blocks.6.proj.bias"]
    torch.save(short, tmp_path / "logs" / "checkpoints" / "slowfast-HTAH" / "acc0.950_e1.ckpt")
    with pytest.raises(RuntimeError, match="missing"):
        small_trainer(small_cfg(tmp_path, name="slowfast-LHand"))


def test_oracle_written_checkpoint_round_trips(tmp_path):
    """a checkpoint the REFERENCE-side module wrote (torch.save(model.state_dict())) loads strict=True and gives the
    oracle's logits; the product's own checkpoint loads back into the oracle module strict=True."""
    cfg = small_cfg(tmp_path)
    torch.manual_seed(3)
    om = o.init_my_slowfast(7, (5, 15), (64, 8), depth=18)
    ckdir = tmp_path / "logs" / "checkpoints" / "slowfast-Torso"
    ckdir.mkdir(parents=True)
    torch.save(om.state_dict(), ckdir / "acc0.500_e3.ckpt")
    t = small_trainer(cfg)
    clips = torch.randn(2, 4, 21, 64, 64, generator=torch.Generator().manual_seed(1))
    om.eval(); t.model.eval()
    with torch.no_grad():
        want = om(o.prepare_slowfast_data(clips))
    x, _ = t.mm.prepare_data({"CropLHand": clips, "label": torch.tensor([0, 1])})
    assert rel_err(t.model(x), want) < 1e-4
    t.save_ckpt(epoch=4, acc=0.75)
    om2 = o.init_my_slowfast(7, (5, 15), (64, 8), depth=18)
    om2.load_state_dict(torch.load(ckdir / "acc0.750_e4.ckpt", weights_only=True), strict=True)
    for (k, a), (_, b) in zip(om.state_dict().items(), om2.state_dict().items()):
        assert torch.equal(a, b), k


# ------------------------------------------------------------------ a2: Trainer.train
def test_trainer_train_best_accuracy_bookkeeping(tmp_path, capsys):
    cfg = small_cfg(tmp_path)
    cfg.MODEL.MAX_EPOCH = 3
    t = small_trainer(cfg)
    real, accs, calls = t.run_eval, iter([0.5, 0.25, 0.75]), []

    def run_eval(loader=None):                          # the real eval runs; only its accuracy is scripted
        r = real(loader)
        assert set(r) == {"ps", "t", "acc", "sv"} and r["ps"].shape[1] == 7
        r["acc"] = next(accs)
        calls.append(r["acc"])
        return r
    t.run_eval = run_eval
    p0 = t.model.engine.P.data.clone()
    t.train()
    out = capsys.readouterr().out
    assert calls == [0.5, 0.25, 0.75]                    # eval after EVERY epoch (train.py:273)
    ckdir = tmp_path / "logs" / "checkpoints" / "slowfast-Torso"
    # epoch 0 improves on 0.0 -> saved; epoch 1 does not -> "Not saved"; epoch 2 improves -> saved; then the final save
    # of train.py:284 rewrites the same name
    assert sorted(os.listdir(ckdir)) == ["acc0.500_e0.ckpt", "acc0.750_e2.ckpt"]
    assert "Not saved. Current best acc: 0.500" in out and out.count("Checkpoint saved") == 3
    assert t.max_historical_acc == 0.75
    assert not torch.equal(p0, t.model.engine.P.data)
    assert int(t.model.engine.adam_step[0]) == 3 * 2    # 4 videos / batch 2 = 2 steps per epoch, drop_last
    # DEBUG: 3 epochs, one batch each, nothing written (train.py:191-195,244-245,257-260)
    cfg2 = small_cfg(tmp_path / "dbg")
    cfg2.DEBUG = True
    t2 = small_trainer(cfg2)
    t2.train()
    assert "Ignore checkpoint saving under debug mode" in capsys.readouterr().out
    assert not (tmp_path / "dbg" / "logs" / "checkpoints" / "slowfast-Torso").exists() or \
        os.listdir(tmp_path / "dbg" / "logs" / "checkpoints" / "slowfast-Torso") == []
    assert int(t2.model.engine.adam_step[0]) == 3


# ------------------------------------------------------------------ yaml surface
REF_YAML = {   # the key sets and value spellings of /root/reference/config/*.yaml (data, not text)
    "slowfast-Torso": "CHALEARN:\n  BATCH_SIZE: 55\n\nMODEL:\n  NAME: 'slowfast-Torso' \n  R3D_INPUT: 'CropTorso'\n  LR: 2e-4\n  MAX_EPOCH: 50\n",
    "slowfast-LHand": "CHALEARN:\n  BATCH_SIZE: 300\n\nMODEL:\n  NAME: 'slowfast-LHand' \n  R3D_INPUT: 'CropLHand'\n  LR: 2e-4\n  MAX_EPOCH: 50",
    "res2d": "CHALEARN:\n  BATCH_SIZE: 60\n  CLIP_LEN: 10\n\nMODEL:\n  NAME: 'res2d' \n  LR: 5e-4\n  MAX_EPOCH: 400",
    "res3d": "CHALEARN:\n  BATCH_SIZE: 30\n\nMODEL:\n  NAME: 'res3d' ",
}


def test_yaml_merge_like_the_reference_main(tmp_path, monkeypatch):
    for name, text in REF_YAML.items():
        (tmp_path / f"{name}.yaml").write_text(text)
    cfg = get_cfg()
    assert (cfg.CHALEARN.BATCH_SIZE, cfg.MODEL.LR, cfg.MODEL.R3D_INPUT, cfg.CHALEARN.CLIP_LEN) == (10, 5e-4, "CropHTAH", 20)
    cfg.merge_from_file(tmp_path / "slowfast-Torso.yaml")
    assert cfg.CHALEARN.BATCH_SIZE == 55 and cfg.MODEL.NAME == "slowfast-Torso" and cfg.MODEL.R3D_INPUT == "CropTorso"
    assert cfg.MODEL.LR == 2e-4 and isinstance(cfg.MODEL.LR, float)      # PyYAML reads '2e-4' as a str; yacs casts
    assert cfg.MODEL.MAX_EPOCH == 50 and cfg.CHALEARN.NUM_CLASS == 249
    # train.py:406-410 mutates ONE cfg cumulatively over yaml_list: keys a later file does not name keep the earlier value
    cfg.merge_from_file(tmp_path / "res2d.yaml")
    assert cfg.MODEL.NAME == "res2d" and cfg.CHALEARN.CLIP_LEN == 10 and cfg.MODEL.R3D_INPUT == "CropTorso"
    assert cfg.MODEL.LR == 5e-4
    fresh = get_cfg()
    assert fresh.MODEL.NAME == "new_feature_test"                         # get_cfg() hands out clones
    # ../cfg_override.yaml, one directory ABOVE the cwd (config/defaults.py:56-61, train.py:411-413)
    from video_classification_amd.config import get_override_cfg
    (tmp_path / "cfg_override.yaml").write_text("CHALEARN:\n  ROOT: '/data/iso'\n")
    work = tmp_path / "repo"
    work.mkdir()
    monkeypatch.chdir(work)
    assert get_override_cfg().CHALEARN.ROOT == "/data/iso"
    with pytest.raises(KeyError):
        bad = tmp_path / "bad.yaml"
        bad.write_text("MODEL:\n  NOPE: 1\n")
        get_cfg().merge_from_file(bad)
    with pytest.raises(ValueError):
        bad.write_text("CHALEARN:\n  BATCH_SIZE: 'many'\n")
        get_cfg().merge_from_file(bad)


# ------------------------------------------------------------------ MODEL.FUSE = False, MODEL.ARCH
def test_model_fuse_false_matches_the_oracle(tmp_path):
    """my_slowfast.py:90-92: fusion modules are Identity, the slow stages see 64/256/512/1024 input channels"""
    from test_engine_cpu import engine_grads_as_state_dict, oracle_train_step_with_engine_mask, randomize
    from helpers import rel_l2
    cfg = small_cfg(tmp_path)
    cfg.MODEL.FUSE = False
    torch.manual_seed(5)
    om = o.init_my_slowfast(7, (5, 15), (64, 8), fuse=False, depth=18)
    randomize(om, 4)
    m = ModelManager(cfg, device="cpu", backend=EmuBackend()).init_model()
    assert set(m.state_dict()) == set(om.state_dict()) and not any("fusion" in k for k in m.state_dict())
    assert tuple(m.state_dict()["blocks.1.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight"].shape) == (64, 64, 1, 1, 1)
    m.load_state_dict(om.state_dict(), strict=True)
    clips = torch.randn(2, 4, 21, 64, 64, generator=torch.Generator().manual_seed(2))
    x = o.prepare_slowfast_data(clips)
    om.eval(); m.eval()
    with torch.no_grad():
        assert rel_err(m(list(x)), om(list(x))) < 1e-4
    m.train()
    labels = torch.tensor([1, 4])
    y_o, _ = oracle_train_step_with_engine_mask(om, m.engine, x, labels)
    y_m = m(list(x))
    torch.nn.functional.cross_entropy(y_m, labels).backward()
    assert rel_err(y_m.detach(), y_o) < 1e-4
    gsd = engine_grads_as_state_dict(m.engine)
    for k, p in om.named_parameters():
        assert rel_l2(gsd[k], p.grad) < 5e-2, k


def test_model_arch_canonical8x8_through_the_config_surface(tmp_path):
    """MODEL.ARCH = canonical8x8: the metric's model reachable from yaml -- BGR frames, PackPathway inside the stem."""
    from test_engine_cpu import randomize
    cfg = small_cfg(tmp_path, clip_len=8)
    cfg.MODEL.ARCH = "canonical8x8"
    mm = ModelManager(cfg, device="cpu", backend=EmuBackend())
    m = mm.init_model()
    assert m.slow_t_index.tolist() == [0, 7] and m.spec.head_pool_kernels == ((2, 2, 2), (8, 2, 2))
    torch.manual_seed(9)
    om = o.mini_slowfast(7, ref_style=False, depth=18)
    randomize(om, 6)
    m.load_state_dict(om.state_dict(), strict=True)
    clips = torch.randn(2, 8, 21, 64, 64, generator=torch.Generator().manual_seed(4))
    x, y = mm.prepare_data({"CropLHand": clips, "label": torch.tensor([3, 0])})
    assert x[0].shape == (2, 3, 8, 64, 64) and x[0].data_ptr() == clips.data_ptr()       # a view of the batch memory
    om.eval(); m.eval()
    with torch.no_grad():
        want = om(o.pack_pathway(clips.permute(0, 2, 1, 3, 4)[:, 0:3], alpha=4))
    assert rel_err(m(x), want) < 1e-4
    # and through the Trainer: the fused step picks up model.slow_t_index
    cfg.DEBUG = True
    t = small_trainer(cfg)
    t.train_epoch()
    assert int(t.model.engine.adam_step[0]) == 1
    with pytest.raises(ValueError):
        bad = small_cfg(tmp_path)
        bad.MODEL.ARCH = "x3d"
        ModelManager(bad, device="cpu", backend=EmuBackend())


# ------------------------------------------------------------------ res2d (BASELINE.json config 1): CPU plumbing
def test_res2d_loader_model_loss_plumbing(tmp_path):
    from video_classification_amd.res2d import resnet50_2d
    assert sum(p.numel() for p in resnet50_2d(3, 1000).parameters()) == 25_557_032       # torchvision resnet50
    keys = set(resnet50_2d(50).state_dict())
    assert {"conv1.weight", "layer1.0.downsample.0.weight", "layer4.2.bn3.running_var", "fc.bias"} <= keys
    (tmp_path / "res2d.yaml").write_text(REF_YAML["res2d"])
    cfg = get_cfg()
    cfg.merge_from_file(tmp_path / "res2d.yaml")
    cfg.CHALEARN.ROOT = str(tmp_path)
    cfg.CHALEARN.BATCH_SIZE = 2                            # BASELINE config 1: batch = 2
    cfg.MODEL.R3D_INPUT = "CropLHand"
    cfg.NUM_CPU = 0
    cfg.DEBUG = True
    t = small_trainer(cfg)
    batch = next(iter(t.train_loader))
    x, y = t.mm.prepare_data(batch)
    assert tuple(x.shape) == (2, 50, 64, 64) and tuple(y.shape) == (2,)      # [:, :, :5] -> (N, T*C, H, W), T = 10
    assert torch.equal(x[:, 5:10], batch["CropLHand"][:, 1, :5])             # frame-major channel stacking
    assert tuple(t.model.conv1.weight.shape) == (64, 50, 7, 7)
    loss_avg, acc = t.train_epoch()
    assert np.isfinite(loss_avg) and t.step.steps == 1
    r = t.run_eval()
    assert r["ps"].shape[1] == 1000 and np.allclose(r["ps"].sum(1), 1.0, atol=1e-5) and len(r["sv"]) == 3
