"""Generate tests/golden/aux_*.npz from the REFERENCE's own code for the rows either side of the hot path:

  aux_sparse_model.npz   SparseModel forward + CrossEntropyLoss gradients      (train_sparse.py:88-104,157-158,171-176)
  aux_sparse_test.npz    SparseTrainer.test's per-video aggregation + accuracy   (train_sparse.py:197-240)
  aux_run_eval.npz       Trainer.run_eval: batching, numpy softmax, per-video mean, argmax, accuracy (train.py:287-370)

Run in the build container only (needs /root/reference):   python tests/golden/make_aux_golden.py

train.py / train_sparse.py import packages that are absent here (torchvision, pytorchvideo, yacs via config.defaults,
cv2 via the dataset module, turtle); those imports are replaced by empty stub modules -- ordinary ModuleNotFoundErrors,
nothing was refused.  The reference methods run UNMODIFIED on objects created with object.__new__ (their __init__ builds
CUDA models and reads the dataset): the attributes the methods read are set by hand, and Tensor.cuda is made the
identity for the duration (there is no GPU in this container).  Data only -- no reference source text is stored.
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


def import_reference():
    anything = lambda *a, **k: None  # noqa: E731

    def set_attributes(self, params=None):
        if params:
            for k, v in params.items():
                if k != "self":
                    setattr(self, k, v)
    _stub("turtle", forward=anything)
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
    _stub("config.defaults", get_cfg=anything, get_override_cfg=anything)
    for opt in ("tqdm", "requests", "matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image"):
        try:
            importlib.import_module(opt)
        except Exception:
            _stub(opt, Image=object)
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        train = importlib.import_module("train")
        train_sparse = importlib.import_module("train_sparse")
    finally:
        os.chdir(cwd)
    return train, train_sparse


def gold_sparse_model(ts):
    torch.manual_seed(21)
    C, P, N = 7, 3, 10
    m = ts.SparseModel(C, P)
    x = torch.randn(N, P, C)
    t = torch.randint(0, C, (N,))
    y = m(x)
    loss = torch.nn.CrossEntropyLoss()(y, t)
    loss.backward()
    rec = {"x": x.numpy(), "t": t.numpy(), "y": y.detach().numpy(), "loss": np.float32(loss.item())}
    for k, v in m.state_dict().items():
        rec["state/" + k] = v.numpy()
    for k, p in m.named_parameters():
        rec["grad/" + k] = p.grad.numpy()
    # one Adam(1e-3) step of the reference loop (train_sparse.py:157,171-176)
    m2 = ts.SparseModel(C, P)
    m2.load_state_dict(m.state_dict())
    opt = torch.optim.Adam(m2.parameters(), lr=1e-3)
    for _ in range(3):
        l2 = torch.nn.CrossEntropyLoss()(m2(x), t)
        opt.zero_grad()
        l2.backward()
        opt.step()
    for k, v in m2.state_dict().items():
        rec["after3/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "aux_sparse_model.npz"), **rec)
    print("sparse model", {k: v.shape for k, v in rec.items() if k.startswith("state/")}.__len__(), "tensors")


def gold_sparse_test(ts):
    torch.manual_seed(22)
    C, P = 6, 3
    sv = [3, 1, 4, 2, 5]
    labels = np.repeat(np.array([2, 0, 5, 1, 3]), sv)
    N = int(sum(sv))
    ps = torch.randn(N, P, C)
    ps[torch.arange(N), :, torch.from_numpy(labels)] += 1.0     # part scores that carry some signal
    tr = object.__new__(ts.SparseTrainer)
    tr.sparse_model = ts.SparseModel(C, P)
    with torch.no_grad():
        for fc in tr.sparse_model.fcs:                            # positive fusion weights, as a trained model has
            fc.weight.abs_()
    tr.test_loader = [{"ps": ps[i:i + 4], "t": torch.from_numpy(labels[i:i + 4])} for i in range(0, N, 4)]
    tr.test_dataset = types.SimpleNamespace(sv=np.array(sv))
    tr.max_accuracy = 0.0
    saved = {}
    tr.save_ckpt = lambda acc, epoch: saved.update(acc=acc, epoch=epoch)
    tr.test(epoch=7)
    with torch.no_grad():
        scores = tr.sparse_model(ps).numpy()
    rec = {"ps": ps.numpy(), "labels": labels, "sv": np.array(sv), "scores": scores,
           "accuracy": np.float64(tr.max_accuracy), "saved_epoch": np.int64(saved.get("epoch", -1))}
    for k, v in tr.sparse_model.state_dict().items():
        rec["state/" + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "aux_sparse_test.npz"), **rec)
    print("sparse test accuracy", tr.max_accuracy)


def gold_run_eval(train):
    torch.manual_seed(23)
    C, F = 9, 9
    net = torch.nn.Linear(F, C)
    with torch.no_grad():
        net.weight.add_(torch.eye(C) * 1.5)                       # a model that is right more often than not
    sv = [2, 3, 1, 4, 2, 3, 1]                       # clips per video
    vid_label = [4, 0, 8, 2, 2, 7, 1]
    videos = []
    for nclip, lab in zip(sv, vid_label):
        videos.append([{"feat": torch.randn(F) * 1.2 + 2.0 * torch.nn.functional.one_hot(torch.tensor(lab), F), "label": lab}
                       for _ in range(nclip)])
    loader = [videos[0:2], videos[2:3], videos[3:6], videos[6:7]]     # DataLoader(batch_size=k, collate_fn=lambda x: x)
    tr = object.__new__(train.Trainer)
    tr.mm = types.SimpleNamespace(prepare_data=lambda b: (b["feat"], b["label"]))
    seen = []

    class Recorder(torch.nn.Module):          # keeps the logits exactly as the reference saw them, batch by batch
        def forward(self, x):
            y = net(x)
            seen.append(y.detach().clone())
            return y
    tr.model = Recorder()
    tr.test_loader = loader
    tr.batch_size = 4
    tr.debug = False
    res = tr.run_eval()
    logits = torch.cat(seen, dim=0).numpy()
    rec = {"logits": logits, "labels": np.array([c["label"] for v in videos for c in v]), "sv": np.array(res["sv"]),
           "ps": res["ps"], "t": res["t"], "acc": np.float64(res["acc"]), "batch_size": np.int64(4)}
    np.savez_compressed(os.path.join(OUT, "aux_run_eval.npz"), **rec)
    print("run_eval acc", res["acc"], "ps", res["ps"].shape)


if __name__ == "__main__":
    train_mod, ts_mod = import_reference()
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        gold_sparse_model(ts_mod)
        gold_sparse_test(ts_mod)
        gold_run_eval(train_mod)
    finally:
        torch.Tensor.cuda = orig_cuda
