"""Rows either side of the hot path on the CPU: the oracle restatements against the golden vectors
captured from the reference's own code, and the host logic of the product (score-table layout, state-dict keys,
crop-offset ranges, eval batching through the emulated backend)."""
import os

import numpy as np
import pytest
import torch

from oracle import aux_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_run_eval_restatement_matches_reference_fixture():
    z = np.load(os.path.join(GOLD, "aux_run_eval.npz"))
    ps, correct, acc = aux_ref.run_eval_scores(z["logits"], z["labels"], list(z["sv"]))
    assert np.array_equal(ps, z["ps"])                # same numpy expression on the same build: bit-exact
    assert np.array_equal(z["t"], z["labels"])        # the reference returns labels in clip order
    assert acc == float(z["acc"]) and 0.0 < acc < 1.0


def test_sparse_model_restatement_matches_reference_fixture():
    z = np.load(os.path.join(GOLD, "aux_sparse_model.npz"))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    m = aux_ref.SparseModelRef(x.shape[2], x.shape[1])
    sd = {k[len("state/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("state/")}
    assert set(sd) == set(m.state_dict().keys())
    m.load_state_dict(sd, strict=True)
    y = m(x)
    assert torch.equal(y.detach(), torch.from_numpy(z["y"]))
    loss = torch.nn.CrossEntropyLoss()(y, t)
    assert abs(float(loss) - float(z["loss"])) < 1e-6
    loss.backward()
    for k, p in m.named_parameters():
        assert torch.equal(p.grad, torch.from_numpy(z["grad/" + k])), k


def test_sparse_test_restatement_matches_reference_fixture():
    z = np.load(os.path.join(GOLD, "aux_sparse_test.npz"))
    _, acc = aux_ref.sparse_test_scores(z["scores"], z["labels"], list(z["sv"]))
    assert acc == float(z["accuracy"]) and 0.0 < acc < 1.0


def test_preprocess_restatement_and_lut():
    from video_classification_amd.input_pipeline import draw_crop_offsets, normalize_lut
    lut = normalize_lut()
    u8 = torch.arange(256, dtype=torch.uint8).view(1, 16, 16, 1)
    ref = aux_ref.to_tensor_normalize(u8).reshape(-1)
    assert torch.equal(lut, ref)                      # the table holds exactly what ToTensor+Normalize computes
    assert abs(float(lut[0]) + 2.0) < 1e-6 and abs(float(lut[255]) - (1 - 0.45) / 0.225) < 1e-6
    off = draw_crop_offsets(200, 12, torch.Generator().manual_seed(0))
    assert off.dtype == torch.int32 and int(off.min()) == 0 and int(off.max()) == 24
    clip = torch.randn(2, 3, 10, 10)
    c = aux_ref.random_crop(clip, 1, 2, 0)
    assert c.shape == clip.shape and torch.equal(c[..., :-1, 1:], clip[..., 1:, :-1]) and float(c[..., -1, :].abs().max()) == 0


def test_sparse_dataset_layout_and_state_dict_keys(tmp_path):
    import pickle
    from emu_backend import EmuBackend
    from video_classification_amd.sparse import SparseFusionDataset, SparseModel
    rng = np.random.default_rng(0)
    sv, C = [2, 1, 3], 5
    n = sum(sv)
    labels = np.repeat(np.array([1, 4, 0]), sv)
    for name in ("slowfast-RHand", "slowfast-HTAH", "slowfast-LHand"):
        with open(tmp_path / name, "wb") as f:
            pickle.dump({"ps": rng.random((n, C), dtype=np.float32), "t": labels, "acc": 0.5, "sv": sv}, f)
    ds = SparseFusionDataset(tmp_path)
    assert ds.part_names == ["slowfast-HTAH", "slowfast-LHand", "slowfast-RHand"]     # sorted by part name
    assert (ds.num_part, ds.num_N, ds.num_class) == (3, n, C) and len(ds) == n
    item = ds[2]
    assert item["ps"].shape == (3, C) and item["t"] == labels[2]
    m = SparseModel(C, 3, device="cpu", backend=EmuBackend(), seed=1)
    sd = m.state_dict()
    assert set(sd) == {f"fcs.{c}.{s}" for c in range(C) for s in ("weight", "bias")}
    assert tuple(sd["fcs.0.weight"].shape) == (1, 3) and tuple(sd["fcs.0.bias"].shape) == (1,)
    ref = aux_ref.SparseModelRef(C, 3)
    ref.load_state_dict(sd, strict=True)              # the reference module accepts the product's checkpoint
    x = torch.randn(4, 3, C)
    assert torch.allclose(m(x), ref(x), atol=1e-6)
    m2 = SparseModel(C, 3, device="cpu", backend=EmuBackend(), seed=2)
    m2.load_state_dict(ref.state_dict())
    assert torch.allclose(m2(x), ref(x), atol=1e-6)
    # a foreign pickle that names anything but numpy's array reconstructors is refused, not executed
    import os
    with open(tmp_path / "slowfast-Evil", "wb") as f:
        pickle.dump({"ps": os.getcwd, "t": labels, "sv": sv}, f)
    with pytest.raises(pickle.UnpicklingError):
        SparseFusionDataset(tmp_path)
