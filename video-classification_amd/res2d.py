"""`MODEL.NAME: res2d` -- host plumbing only (BASELINE.json config 1: "CPU reference path (plumbing, no GPU)").

The reference builds torchvision's ResNet-50 from torch.hub and swaps its first conv for a 50-channel one
(/root/reference/train.py:64-68); its input is the clip's BGR+UV channels with the T frames stacked on the channel axis,
`batch[...][:, :, :5]` reshaped (N, T*C, H, W) (train.py:70-76; config/res2d.yaml: CLIP_LEN 10 -> 50 channels).
This is NOT a video path and not on the accelerated hot path: it exists so
that `config/res2d.yaml` runs loader -> prepare_data -> model -> loss -> Adam end to end.  The network is a plain
torch.nn ResNet-50 with torchvision's module / state-dict names (conv1, bn1, layer{1..4}.{i}.conv{1,2,3} / bn{1,2,3} /
downsample.{0,1}, fc), so a torchvision checkpoint loads; torch.hub and torchvision are unreachable offline.
It never touches libsfk and the SlowFast path never touches it.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)      # torchvision v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet2d(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), in_channels: int = 50, num_classes: int = 1000):
        super().__init__()
        self.inplanes = 64
        # train.py:66: model.conv1 = Conv2d(50, 64, kernel_size=(7,7), stride=(2,2), padding=(3,3), bias=False)
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * 4, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes: int, blocks: int, stride: int) -> nn.Sequential:
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def resnet50_2d(in_channels: int = 50, num_classes: int = 1000) -> ResNet2d:
    """torchvision `resnet50` (25,557,032 parameters at 3 input channels / 1000 classes) with the reference's conv1."""
    return ResNet2d((3, 4, 6, 3), in_channels, num_classes)


class TorchStep:
    """The reference's five hot lines (train.py:225-231) as they are, for the one model that is a plain torch module;
    same meters as train.TrainStep so Trainer.train_epoch reads them the same way."""

    def __init__(self, model: nn.Module, lr: float):
        self.model = model
        self.optim = torch.optim.Adam(model.parameters(), lr=lr)
        self.criterion = nn.CrossEntropyLoss()
        dev = next(model.parameters()).device
        self.loss = torch.zeros(1, device=dev)
        self.loss_sum = torch.zeros(1, device=dev)
        self.correct = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps = 0

    def reset_meters(self):
        self.loss_sum.zero_()
        self.correct.zero_()
        self.steps = 0

    def __call__(self, x, _unused, y_true, slow_t_index=None):
        self.model.train()
        y_pred = self.model(x)
        loss = self.criterion(y_pred, y_true)
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        with torch.no_grad():
            self.loss[0] = loss.detach()
            self.loss_sum += loss.detach()
            self.correct += (y_pred.argmax(-1) == y_true).sum().to(torch.int32)
        self.steps += 1
        return self.loss


def aggregate_scores_host(logits: torch.Tensor, labels: torch.Tensor, samples_per_video):
    """Trainer.run_eval's numpy post-processing (train.py:337-362) for the torch-module model: softmax, per-video mean,
    argmax against the label of the video's first clip; videos without clips are skipped."""
    ps = torch.softmax(logits.float(), dim=-1)
    pred, correct, base = [], 0, 0
    for s_ in samples_per_video:
        if s_ > 0:
            p = int(ps[base:base + s_].mean(0).argmax())
            correct += int(p == int(labels[base]))
            pred.append(p)
        else:
            pred.append(-1)
        base += s_
    return ps, torch.tensor(pred, dtype=torch.int32), correct
