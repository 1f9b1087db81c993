"""GPU: tf_image_prepare / tinyfaces.datasets.augment against the oracle restatement and the
reference goldens.  Integer stage (Pillow's 8-bit resample, crop, paste, flip) must be bit-exact; so must the float stage
(x / 255, (x - mean) / std evaluated in float32 like torch's ToTensor + Normalize)."""
import numpy as np
import pytest
import torch

from gpu_util import report
from test_oracle_augment import synth_image

pytestmark = pytest.mark.gpu


def _u8_from_normalized(x, mean, std):
    """invert Normalize(ToTensor(u8)) exactly enough to recover the uint8 image (values are k/255)."""
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    return np.rint((x * s + m) * 255).astype(np.int64).transpose(1, 2, 0)


@pytest.mark.parametrize("shape", [(37, 53, 18, 26), (37, 53, 74, 106), (100, 100, 100, 57), (64, 48, 129, 48), (33, 77, 500, 389),
                                   (250, 333, 125, 166), (11, 13, 5, 6), (96, 96, 31, 200), (480, 640, 240, 320), (480, 640, 960, 1280),
                                   (90, 120, 22, 30)])
def test_resize_normalize_equals_oracle_bitwise(shape):
    from oracle import augment as oa
    from tinyfaces import ops
    H, W, oh, ow = shape
    img = synth_image(H * 7 + W, H, W)
    ref_u8 = oa.pil_resize_u8(img, oh, ow)
    ref = oa.to_normalized_tensor(ref_u8)
    got = ops.image_prepare(torch.from_numpy(img).cuda(), resized_hw=(oh, ow)).cpu().numpy()
    u8 = _u8_from_normalized(got, ops.IMAGE_MEAN, ops.IMAGE_STD)
    nd = int((u8 != ref_u8).sum())
    report(f"image_prepare_resize[{shape}]", u8_mismatches=nd, float_bitwise=bool(np.array_equal(got, ref)))
    assert nd == 0
    assert np.array_equal(got, ref)                      # float32 bit pattern, not a tolerance


def test_crop_paste_flip_window_only():
    """A window of the resized image pasted off-centre on the mean colour, mirrored; compared with doing the same on the
    oracle's full resize (the kernel never resamples outside the window)."""
    from oracle import augment as oa
    from tinyfaces import ops
    img = synth_image(77, 300, 420)
    full = oa.pil_resize_u8(img, 600, 840)
    cy, cx, ch, cw, py, px = 37, 111, 400, 333, 60, 90
    for flip in (False, True):
        buf = np.zeros((500, 500, 3), np.uint8)
        buf[:] = (123, 116, 103)
        buf[py:py + ch, px:px + cw] = full[cy:cy + ch, cx:cx + cw]
        if flip:
            buf = np.fliplr(buf).copy()
        ref = oa.to_normalized_tensor(buf)
        got = ops.image_prepare(torch.from_numpy(img).cuda(), resized_hw=(600, 840), crop=(cy, cx, ch, cw), paste=(py, px), flip=flip,
                                out_hw=(500, 500)).cpu().numpy()
        assert np.array_equal(got, ref), flip


def test_process_inputs_device_vs_reference_golden(golden):
    """The whole training augmentation with the pixels on the GPU: same seeded np.random decisions, then the tensor that
    main.py:44-46 would build from the reference's process_inputs output, bit for bit, and the same surviving boxes."""
    from oracle import augment as oa
    from tinyfaces.datasets import augment as da
    g = golden("augment")
    for n, (seed, H, W) in enumerate(g["cases"].tolist()):
        img = synth_image(seed, H, W)
        rng = np.random.RandomState(1000 + seed)
        x, boxes, paste, flip = da.process_inputs(torch.from_numpy(img).cuda(), g[f"a{n}_boxes_in"].copy(), rng=rng)
        ref = oa.to_normalized_tensor(g[f"a{n}_img"])
        assert np.array_equal(x.cpu().numpy(), ref), n
        assert boxes.shape == g[f"a{n}_boxes"].shape and np.array_equal(boxes, g[f"a{n}_boxes"]), n
    report("process_inputs_device", cases=len(g["cases"]))


def test_image_prepare_rejects_bad_windows_and_cpu():
    from tinyfaces import ops
    img = torch.zeros(20, 30, 3, dtype=torch.uint8)
    with pytest.raises(RuntimeError, match="no CPU"):
        ops.image_prepare(img)
    with pytest.raises(RuntimeError):
        ops.image_prepare(img.cuda(), crop=(0, 0, 25, 30))             # window taller than the image
    with pytest.raises(RuntimeError):
        ops.image_prepare(img.cuda(), paste=(5, 0), out_hw=(20, 30))   # pasted window leaves the output


def test_wider_dataloader_end_to_end(tmp_path):
    """get_dataloader on a miniature WIDER tree (datasets/__init__.py:11-52, wider_face.py): decode on the host, augmentation
    and target assignment on the GPU; a batch has the reference's contract and equals the same steps done by hand."""
    from types import SimpleNamespace
    from PIL import Image
    from tinyfaces import transforms
    from tinyfaces.datasets import get_dataloader
    from tinyfaces.datasets import augment as da
    root = tmp_path / "WIDER_train" / "images" / "0--Parade"
    root.mkdir(parents=True)
    ann, imgs = "", {}
    for k, (H, W) in enumerate([(600, 800), (420, 380), (700, 1000)]):
        name = f"0--Parade/img{k}.png"
        arr = synth_image(40 + k, H, W)
        Image.fromarray(arr, "RGB").save(tmp_path / "WIDER_train" / "images" / name)          # PNG: lossless, decode == arr
        imgs[name] = arr
        ann += f"{name}\n2\n{50 + 10 * k} {60 + 5 * k} 120 150 0 0 0 0 0 0\n{200 + k} {180 + k} 40 48 0 0 0 0 0 0\n"
    ann_file = tmp_path / "train.txt"
    ann_file.write_text(ann)
    args = SimpleNamespace(batch_size=3, workers=0, dataset_root=str(tmp_path), debug=False)
    tf = transforms.Compose([transforms.ToTensor(), transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    loader, templates = get_dataloader(ann_file, args, img_transforms=tf, train=False, split="train")
    assert len(loader) == 1 and templates.shape == (25, 5)
    np.random.seed(123)
    (x, cm, rm), = list(loader)
    assert x.shape == (3, 3, 500, 500) and x.dtype == torch.float32 and x.is_cuda
    assert cm.shape == (3, 25, 63, 63) and rm.shape == (3, 100, 63, 63) and cm.is_cuda
    assert set(torch.unique(cm).tolist()) <= {-1.0, 0.0, 1.0} and int((cm == 1).sum()) > 0
    # by hand, same np.random stream
    np.random.seed(123)
    for i, d in enumerate(loader.dataset.data):
        xi, _, _, _ = da.process_inputs(torch.from_numpy(imgs[d["img_path"]]).cuda(), d["bboxes"])
        assert torch.equal(xi, x[i]), i
