"""The three torchvision transforms main.py / evaluate_model.py construct (main.py:44-46,
evaluate_model.py:35-37), provided here because torchvision is not part of this stack.
Host-side image plumbing, outside the hot path."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    """uint8 HWC ndarray / PIL image -> float32 CHW in [0,1]."""

    def __call__(self, pic):
        arr = np.asarray(pic)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
        return t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, t):
        mean = torch.as_tensor(self.mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
        return (t - mean) / std


def to_tensor(pic):
    return ToTensor()(pic)


def to_uint8_chw(t):
    """float CHW tensor -> (x*255) truncated to uint8, still CHW (ndarray): the pixels of to_pil_image(t), planar."""
    a = t.detach().cpu().numpy()
    if t.is_floating_point():
        # pic.mul(255).byte() of torchvision's to_pil_image, in numpy: the same IEEE product and the same truncation, without torch's intra-op thread
        # fan-out (r6: 90 ms per 1280 x 960 image on a 256-thread host against 5 ms -- it was 94 % of get_detections end to end)
        a = (a * a.dtype.type(255)).astype(np.uint8)
    return np.ascontiguousarray(a)


def to_uint8_hwc(t):
    """... interleaved (H, W, 3): what PIL holds."""
    return np.ascontiguousarray(to_uint8_chw(t).transpose(1, 2, 0))


def to_pil_image(t):
    """float CHW tensor -> (x*255) truncated to uint8 -> PIL RGB (what evaluation.py:40 relies on)."""
    from PIL import Image
    return Image.fromarray(to_uint8_hwc(t), mode="RGB")


def resize(img, size):
    """int -> shorter side = size, longer = int(size*long/short), PIL bilinear (evaluation.py:46-47)."""
    from PIL import Image
    w, h = img.size
    if isinstance(size, int):
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = size, int(size * long / short)
        new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        new_h, new_w = size
    return img.resize((new_w, new_h), Image.BILINEAR)
