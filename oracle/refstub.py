"""Container-only tooling: make `/root/reference` importable where torchvision,
pyclust and pyclustering are absent.

`install()` registers `sys.modules` stubs BEFORE `import tinyfaces.*`.  The stub bodies are
this repo's own restatements of third-party torchvision 0.18 behaviour (appendix A.3) --
they are NOT reference code.  Used only by oracle/tools/make_golden.py and by the tests
that re-check the oracle against the live reference when /root/reference exists.
Nothing here runs on the GPU box (`/root/reference` does not exist there).
"""
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = "/root/reference"


# ---- torchvision.transforms restated (appendix A.3) -------------------------------
def to_tensor(pic):
    """uint8 HWC ndarray / PIL -> float32 CHW in [0,1] (divide by 255)."""
    from PIL import Image
    if isinstance(pic, Image.Image):
        arr = np.array(pic, dtype=np.uint8, copy=True)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        return torch.from_numpy(arr).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    arr = np.asarray(pic)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    if t.dtype == torch.uint8:
        return t.to(torch.float32).div(255)
    return t


def to_pil_image(pic, mode=None):
    """float CHW tensor -> (x*255) truncated to uint8 -> PIL RGB."""
    from PIL import Image
    if isinstance(pic, torch.Tensor):
        if pic.is_floating_point():
            pic = pic.mul(255).byte()
        arr = pic.cpu().numpy().transpose(1, 2, 0)
    else:
        arr = np.asarray(pic)
    if arr.shape[2] == 1:
        return Image.fromarray(arr[:, :, 0], mode="L")
    return Image.fromarray(np.ascontiguousarray(arr), mode="RGB")


def resize(img, size, interpolation=None):
    """int size -> shorter side = size, longer = int(size*long/short); (h, w) -> exact.
    PIL bilinear (PIL applies antialiasing support scaling when down-sampling)."""
    from PIL import Image
    w, h = img.size
    if isinstance(size, int):
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long / short)
        new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        new_h, new_w = size
    return img.resize((new_w, new_h), Image.BILINEAR)


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    def __call__(self, pic):
        return to_tensor(pic)


class ToPILImage:
    def __call__(self, pic):
        return to_pil_image(pic)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
        return (t - mean) / std


def _nms_torch(boxes, scores, thr):
    from oracle.nms import nms
    keep = nms(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), thr)
    return torch.from_numpy(keep)


def install(reference_root=REFERENCE_ROOT):
    """Install stubs and put the reference on sys.path.  Idempotent."""
    if "torchvision" not in sys.modules or getattr(sys.modules["torchvision"], "_tf_stub", False) is False:
        from oracle import resnet as _resnet
        tv = types.ModuleType("torchvision")
        tv.__path__ = []
        tv._tf_stub = True
        models = types.ModuleType("torchvision.models")

        class ResNet101_Weights:  # names only (tinyfaces/models/model.py:4,14)
            IMAGENET1K_V1 = None

        models.ResNet101_Weights = ResNet101_Weights
        models.resnet101 = _resnet.resnet101
        ops = types.ModuleType("torchvision.ops")
        ops.nms = _nms_torch
        tfm = types.ModuleType("torchvision.transforms")
        tfm.__path__ = []
        func = types.ModuleType("torchvision.transforms.functional")
        func.to_tensor, func.to_pil_image, func.resize = to_tensor, to_pil_image, resize
        tfm.functional = func
        tfm.Compose, tfm.ToTensor, tfm.Normalize, tfm.ToPILImage = Compose, ToTensor, Normalize, ToPILImage
        tv.models, tv.ops, tv.transforms = models, ops, tfm
        sys.modules.update({
            "torchvision": tv, "torchvision.models": models, "torchvision.ops": ops,
            "torchvision.transforms": tfm, "torchvision.transforms.functional": func})
    if "pyclust" not in sys.modules:
        pc = types.ModuleType("pyclust")
        pc.KMedoids = object
        sys.modules["pyclust"] = pc
        p0 = types.ModuleType("pyclustering"); p0.__path__ = []
        p1 = types.ModuleType("pyclustering.cluster"); p1.__path__ = []
        p2 = types.ModuleType("pyclustering.cluster.kmedoids")
        p2.kmedoids = object
        sys.modules.update({"pyclustering": p0, "pyclustering.cluster": p1,
                            "pyclustering.cluster.kmedoids": p2})
    if reference_root not in sys.path:
        sys.path.insert(0, reference_root)


def available(reference_root=REFERENCE_ROOT):
    import os
    return os.path.isdir(os.path.join(reference_root, "tinyfaces"))
