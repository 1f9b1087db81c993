"""Cost of the training augmentation on the GPU: 12 images (1024x768 uint8, resident) -> 12 x (3, 500, 500) fp32, per scale branch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")]
from tinyfaces import ops
imgs = [torch.randint(0, 256, (768, 1024, 3), dtype=torch.uint8, device="cuda") for _ in range(12)]
out = torch.empty(12, 3, 500, 500, device="cuda")
for name, rhw in (("x1", (768, 1024)), ("x0.5", (384, 512)), ("x2", (1536, 2048))):
    ch, cw = min(500, rhw[0] - 3), min(500, rhw[1] - 5)
    def run():
        for i, im in enumerate(imgs):
            ops.image_prepare(im, resized_hw=rhw, crop=(3, 5, ch, cw), paste=(0, 0), flip=i & 1, out_hw=(500, 500), out=out[i])
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): run()
    b.record(); torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 20 * 1e3:.1f} us per batch of 12 ({a.elapsed_time(b) / 240 * 1e3:.1f} us per image)")
