"""Oracle: torch-CPU fp32 restatement of the detector network.  Test infrastructure only.

Restates tinyfaces/models/model.py:12-128 (`DetectionModel`): ResNet-101 trunk minus
layer4 (trunk = oracle/resnet.py, a restatement of third-party torchvision -- PARITY
UNPINNED there), 1x1 heads score_res3/score_res4, frozen bilinear ConvTranspose2d, crop +
add.  The head / upsample-init / crop logic (reference-owned) is pinned by
tests/golden/model_*.npz produced by running the reference's own model.py over the stub
trunk with a shared state_dict.
"""
import numpy as np
import torch
from torch import nn

from oracle.resnet import resnet101


def bilinear_kernel(k=4):
    """model.py:45-63: outer product of [.25,.75,.75,.25] for k=4."""
    factor = np.floor((k + 1) / 2)
    center = factor if k % 2 == 1 else factor + 0.5
    c = np.arange(1, k + 1)
    v = np.ones((1, k)) - (np.abs(c - center) / factor)
    return v.T @ v


class OracleDetectionModel(nn.Module):
    def __init__(self, num_templates=1, num_objects=1):
        super().__init__()
        output = (num_objects + 4) * num_templates                       # model.py:19
        self.model = resnet101()
        del self.model.layer4                                            # model.py:23
        self.score_res3 = nn.Conv2d(512, output, 1)
        self.score_res4 = nn.Conv2d(1024, output, 1)
        self.score4_upsample = nn.ConvTranspose2d(output, output, 4, stride=2, padding=1, bias=False)
        f = np.zeros((output, output, 4, 4))
        f[np.arange(output), np.arange(output)] = bilinear_kernel(4)
        self.score4_upsample.weight = nn.Parameter(torch.Tensor(f))

    def learnable_parameters(self, lr):                                  # model.py:67-87
        return [{"params": self.model.parameters(), "lr": lr},
                {"params": self.score_res3.parameters(), "lr": 0.1 * lr},
                {"params": self.score_res4.parameters(), "lr": 1 * lr},
                {"params": self.score4_upsample.parameters(), "lr": 0}]

    def forward(self, x):                                                # model.py:89-128
        m = self.model
        x = m.maxpool(m.relu(m.bn1(m.conv1(x))))
        x = m.layer1(x)
        res3 = m.layer2(x)
        res4 = m.layer3(res3)
        s3 = self.score_res3(res3)
        s4 = self.score4_upsample(self.score_res4(res4))
        # train (:124) and eval (:112-121) branches select the same top-left slice
        s4 = s4[:, :, 0:s3.size(2), 0:s3.size(3)]
        return s3 + s4


def tame_init_(model, seed=0):
    """Deterministic 'tamed' init used by fixtures and benches:
    plain kaiming init saturates the sigmoid, so every bn3.weight is set to 0.1 (keeps the
    30-block residual stream O(1)) and the head weights are scaled x0.05.  Same recipe on
    every host (torch CPU generator), so no weights need to be shipped."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.startswith("score4_upsample"):
                continue
            if p.dim() == 4:
                fan_out = p.shape[0] * p.shape[2] * p.shape[3]
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / fan_out) ** 0.5)
            elif p.dim() == 2:                      # dead model.fc (model.py:23 keeps it): deterministic too
                p.copy_(0.01 * torch.randn(p.shape, generator=g))
            elif name.endswith("bn3.weight"):
                p.fill_(0.1)
            elif name.endswith(".weight") and p.dim() == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif p.dim() == 1:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
        for name in ("score_res3", "score_res4"):
            getattr(model, name).weight.mul_(0.05)
        for name, b in model.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
    return model
