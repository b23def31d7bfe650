"""GPU parity of the rows either side of the hot path, through the C ABI, against the golden vectors
captured from the reference's own code (tests/golden/aux_*.npz) and the oracle restatements (oracle/aux_ref.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import aux_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def hip():
    from video_classification_amd._lib import HipBackend
    return HipBackend()


def stream():
    return torch.cuda.current_stream().cuda_stream


def test_eval_aggregate_matches_reference_run_eval(hip):
    from video_classification_amd.train import aggregate_scores
    z = np.load(os.path.join(GOLD, "aux_run_eval.npz"))
    logits = torch.from_numpy(z["logits"]).to(DEV)
    labels = torch.from_numpy(z["labels"]).to(DEV)
    ps, pred, ncorrect = aggregate_scores(hip, logits, labels, [int(s) for s in z["sv"]], softmax=True)
    assert np.allclose(ps.cpu().numpy(), z["ps"], rtol=2e-6, atol=1e-8)      # fp32 exp of two libraries
    assert ncorrect / len(z["sv"]) == float(z["acc"])
    _, correct, _ = aux_ref.run_eval_scores(z["logits"], z["labels"], list(z["sv"]))
    want = [int(np.argmax(np.mean(z["ps"][a:b], axis=0))) for a, b in zip(np.cumsum([0] + list(z["sv"]))[:-1], np.cumsum(z["sv"]))]
    assert pred.cpu().tolist() == want and int(sum(correct)) == ncorrect


def test_eval_aggregate_large_and_edge_cases(hip):
    from video_classification_amd.train import aggregate_scores
    gen = torch.Generator().manual_seed(5)
    c = 249
    sv = [int(v) for v in torch.randint(0, 9, (300,), generator=gen)]       # includes videos without clips
    sv[0], sv[-1] = 0, 0
    n = sum(sv)
    labels = torch.repeat_interleave(torch.randint(0, c, (300,), generator=gen), torch.tensor(sv))
    logits = torch.randn(n, c, generator=gen) * 3
    logits[torch.arange(n), labels] += 2.5
    ps, pred, ncorrect = aggregate_scores(hip, logits.to(DEV), labels.to(DEV), sv, softmax=True)
    want_ps = torch.softmax(logits.double(), 1)
    assert torch.allclose(ps.cpu().double(), want_ps, rtol=1e-5, atol=1e-9)
    read, want_pred, want_correct = 0, [], 0
    for s in sv:
        if s == 0:
            want_pred.append(-1)
            continue
        k = int(torch.argmax(ps.cpu()[read:read + s].sum(0) / s))
        want_pred.append(k)
        want_correct += int(k == int(labels[read]))
        read += s
    assert pred.cpu().tolist() == want_pred and ncorrect == want_correct
    # ties go to the first maximum (numpy argmax); scores pass through untouched without softmax
    flat = torch.zeros(3, 7)
    flat[:, 2] = 1.0
    flat[:, 5] = 1.0
    ps2, pred2, _ = aggregate_scores(hip, flat.to(DEV), torch.tensor([2, 2, 2]).to(DEV), [3], softmax=False)
    assert pred2.cpu().tolist() == [2] and torch.equal(ps2.cpu(), flat)


def test_sparse_model_matches_reference_fixture(hip):
    from video_classification_amd.sparse import SparseModel
    z = np.load(os.path.join(GOLD, "aux_sparse_model.npz"))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    C, P = x.shape[2], x.shape[1]
    m = SparseModel(C, P, device=DEV, backend=hip)
    m.load_state_dict({k[len("state/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state/")})
    y = m(x)
    assert np.allclose(y.cpu().numpy(), z["y"], rtol=1e-6, atol=1e-6)
    for _ in range(3):                                    # three Adam(1e-3) steps of the reference loop
        loss, _ = m.train_step(x, t)
    torch.cuda.synchronize()
    sd = m.state_dict()
    for k in sd:
        assert np.allclose(sd[k].cpu().numpy(), z["after3/" + k], rtol=1e-5, atol=2e-6), k
    # gradients of the first step
    m2 = SparseModel(C, P, device=DEV, backend=hip)
    m2.load_state_dict({k[len("state/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state/")})
    loss, _ = m2.train_step(x, t, lr=0.0)
    assert abs(float(loss[0]) - float(z["loss"])) < 1e-6
    g = m2.G.cpu()
    for c in range(C):
        assert np.allclose(g[c * P:(c + 1) * P].numpy(), z[f"grad/fcs.{c}.weight"].reshape(-1), rtol=1e-5, atol=1e-7)
        assert np.allclose(g[C * P + c].numpy(), z[f"grad/fcs.{c}.bias"].reshape(()), rtol=1e-5, atol=1e-7)


def test_sparse_trainer_test_matches_reference_fixture(hip, tmp_path):
    import pickle
    from video_classification_amd.config import get_cfg
    from video_classification_amd.sparse import SparseTrainer
    z = np.load(os.path.join(GOLD, "aux_sparse_test.npz"))
    ps, labels, sv = z["ps"], z["labels"], [int(s) for s in z["sv"]]       # ps: (sample, part, class)
    for split in ("train", "test"):
        d = tmp_path / "logs" / "sparse_fusion" / split
        d.mkdir(parents=True)
        for p in range(ps.shape[1]):
            with open(d / f"part{p}", "wb") as f:
                pickle.dump({"ps": ps[:, p], "t": labels, "acc": 0.0, "sv": sv}, f)
    cfg = get_cfg()
    cfg.CHALEARN.ROOT = str(tmp_path)
    cfg.MODEL.LOGS = "logs"
    tr = SparseTrainer(cfg, device=DEV, backend=hip, batch_size=4)
    tr.sparse_model.load_state_dict({k[len("state/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state/")})
    acc = tr.test(epoch=7)
    assert acc == float(z["accuracy"])
    assert (tmp_path / "logs" / "sparse_fusion_ckpt" / ("acc-%.3f-epoch-7" % acc)).exists()
    tr.train(epochs=2, test_every=1)                       # the loop runs and keeps the best accuracy
    assert tr.max_accuracy >= acc


@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_u8_normalize_crop_bit_exact(hip, out_dtype):
    from video_classification_amd.input_pipeline import DevicePreprocess, draw_crop_offsets
    gen = torch.Generator().manual_seed(6)
    n, t, s, c = 3, 4, 40, 21
    u8 = torch.randint(0, 256, (n, t, s, s, c), generator=gen, dtype=torch.uint8)
    pre = DevicePreprocess(DEV, hip, out_dtype)
    # no augmentation (test clips): ToTensor + Normalize, bit for bit
    got = pre(u8).cpu()
    want = torch.stack([aux_ref.to_tensor_normalize(u8[i]) for i in range(n)])
    assert got.shape == (n, t, c, s, s) and torch.equal(got, want.to(out_dtype))
    # RandomCrop(size, padding = size // 10): one offset per clip, zeros of the normalised tensor outside
    crop = draw_crop_offsets(n, s // 10, gen)
    crop[0] = torch.tensor([0, 8])                          # both extremes occur
    got = pre(u8, crop).cpu()
    want = torch.stack([aux_ref.random_crop(aux_ref.to_tensor_normalize(u8[i]), s // 10, int(crop[i, 0]), int(crop[i, 1]))
                        for i in range(n)])
    assert torch.equal(got, want.to(out_dtype))
    # a ragged frame size (rows that are not 16-byte multiples)
    u8b = torch.randint(0, 256, (1, 2, 13, 13, 5), generator=gen, dtype=torch.uint8)
    got = pre(u8b, torch.tensor([[2, 0]], dtype=torch.int32), padding=1).cpu()
    want = aux_ref.random_crop(aux_ref.to_tensor_normalize(u8b[0]), 1, 2, 0)[None]
    assert torch.equal(got, want.to(out_dtype))


def test_trainer_uint8_batches_and_device_eval(hip, tmp_path):
    """Trainer end to end on uint8 batches: the device-normalised clip equals the float32 clip the reference's dataset
    would have produced, training steps run, and run_eval's result dict keeps the reference contract."""
    from video_classification_amd.config import get_cfg
    from video_classification_amd.train import SyntheticChalearn, Trainer
    cfg = get_cfg()
    cfg.CHALEARN.ROOT = str(tmp_path)
    cfg.CHALEARN.BATCH_SIZE = 2
    cfg.CHALEARN.CLIP_LEN = 4
    cfg.CHALEARN.NUM_CLASS = 7
    cfg.MODEL.R3D_INPUT = "CropLHand"                       # 64 x 64 crops
    cfg.MODEL.NAME = "slowfast-test"
    cfg.DEBUG = True
    tr_set = SyntheticChalearn(cfg, "train", num_videos=4, seed=1, as_uint8=True)
    te_set = SyntheticChalearn(cfg, "test", num_videos=3, clips_per_video=(1, 2), seed=2, as_uint8=True)
    item = tr_set[0]
    assert item["CropLHand_u8"].dtype == torch.uint8 and tuple(item["CropLHand_u8"].shape) == (4, 64, 64, 21)
    loader = torch.utils.data.DataLoader(tr_set, batch_size=2, shuffle=False, drop_last=True)
    tloader = torch.utils.data.DataLoader(te_set, batch_size=2, shuffle=False, collate_fn=lambda x: x)
    trainer = Trainer(cfg, train_loader=loader, test_loader=tloader, device=DEV, backend=hip)
    batch = next(iter(loader))
    x, y = trainer.mm.prepare_data(batch)
    want = torch.stack([aux_ref.random_crop(aux_ref.to_tensor_normalize(batch["CropLHand_u8"][i]), 6,
                                            int(batch["crop"][i, 0]), int(batch["crop"][i, 1])) for i in range(2)])
    assert torch.equal(x[0].cpu(), want.permute(0, 2, 1, 3, 4)[:, 0:5]) and torch.equal(x[1].cpu(), want.permute(0, 2, 1, 3, 4)[:, 5:20])
    trainer.train_epoch()
    res = trainer.run_eval()
    n = sum(te_set.nclips)
    assert res["ps"].shape == (n, 7) and res["t"].shape == (n,) and res["sv"] == te_set.nclips
    assert np.allclose(res["ps"].sum(1), 1.0, atol=1e-5) and 0.0 <= res["acc"] <= 1.0
    _, correct, acc = aux_ref.run_eval_scores(np.log(res["ps"]), res["t"], res["sv"])   # softmax(log p) == p
    assert abs(acc - res["acc"]) < 1e-9


def test_prepare_data_host_to_device_copy_is_asynchronous(tmp_path):
    """row f3 "pinned-memory H2D": with the compute stream busy, prepare_data (float32 batch from PAGEABLE host memory, and
    the uint8 transport) returns while the stream is still running -- the copies are DMAs queued behind the kernels, the
    host does not wait for them (train.py:127's `.cuda()` of a pageable tensor blocks for the whole transfer)."""
    import time
    from video_classification_amd.config import get_cfg
    from video_classification_amd.train import ModelManager
    cfg = get_cfg()
    cfg.MODEL.NAME, cfg.MODEL.R3D_INPUT = "slowfast-Torso", "CropTorso"
    mm = ModelManager(cfg, device=DEV)
    g = torch.Generator().manual_seed(0)
    clips = torch.randn(6, 20, 21, 128, 128, generator=g)                    # 165 MB, pageable
    u8 = torch.randint(0, 256, (6, 20, 128, 128, 21), generator=g, dtype=torch.uint8)
    labels = torch.arange(6)
    assert not clips.is_pinned()
    mm.prepare_data({"CropTorso": clips[:1], "label": labels[:1]})          # warm the allocators / load the kernels
    mm.prepare_data({"CropTorso_u8": u8[:1], "label": labels[:1]})
    torch.cuda.synchronize()
    for batch in ({"CropTorso": clips, "label": labels}, {"CropTorso_u8": u8, "label": labels}):
        torch.cuda._sleep(int(3e9))                                           # ~1 s of busy compute stream
        busy = torch.cuda.Event()
        busy.record()
        t0 = time.perf_counter()
        x, y = mm.prepare_data(batch)
        host_s = time.perf_counter() - t0
        assert not busy.query(), f"prepare_data waited for the compute stream ({host_s:.3f} s)"
        torch.cuda.synchronize()
        assert x[0].shape == (6, 5, 20, 128, 128) and x[1].shape == (6, 15, 20, 128, 128)
        if "CropTorso" in batch:
            assert torch.equal(x[0].cpu(), clips.permute(0, 2, 1, 3, 4)[:, 0:5])
        assert torch.equal(y.cpu(), labels)
