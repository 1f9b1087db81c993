"""CPU: the augmentation / resize restatement (oracle/augment.py) against (1) Pillow itself, (2) the outputs of the reference's
own WIDERFace.process_inputs + DataProcessor.crop_image recorded in tests/golden/augment.npz (oracle/tools/make_golden.py)."""
import numpy as np
import pytest

from oracle import augment


def synth_image(seed, H, W):                      # same generator as oracle/tools/make_golden.py:synth_image
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([127 + 110 * np.sin(xx / (9.0 + c) + c) * np.cos(yy / (6.0 + 2 * c)) for c in range(3)], 2)
    img = img + rng.randint(-2, 3, (H, W, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("shape", [(37, 53, 18, 26), (37, 53, 74, 106), (100, 100, 100, 57), (64, 48, 129, 48), (33, 77, 500, 389),
                                   (250, 333, 125, 166), (11, 13, 5, 6), (96, 96, 31, 200), (301, 203, 150, 101)])
def test_resize_restatement_equals_pillow(shape):
    Image = pytest.importorskip("PIL.Image")
    H, W, oh, ow = shape
    img = synth_image(H * 7 + W, H, W)
    ref = np.array(Image.fromarray(img, "RGB").resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(augment.pil_resize_u8(img, oh, ow), ref)


def test_eval_pyramid_resize_golden(golden):
    g = golden("augment")
    img = synth_image(5, 180, 240)
    for k, s in enumerate((0.5, 2)):
        size = int(180 * s)
        ref = g[f"r{k}_img"]
        assert ref.shape[:2] == (size, int(size * 240 / 180))
        assert np.array_equal(augment.pil_resize_u8(img, ref.shape[0], ref.shape[1]), ref)


def test_process_inputs_vs_reference_golden(golden):
    g = golden("augment")
    seen = set()
    for n, (seed, H, W) in enumerate(g["cases"].tolist()):
        img = synth_image(seed, H, W)
        rng = np.random.RandomState(1000 + seed)                  # the reference ran under np.random.seed(1000 + seed)
        r = augment.process_inputs(img, g[f"a{n}_boxes_in"].copy(), rng=rng)
        assert np.array_equal(r["img"], g[f"a{n}_img"]), (n, r["scale"], r["flip"])
        assert r["bboxes"].shape == g[f"a{n}_boxes"].shape and np.array_equal(r["bboxes"], g[f"a{n}_boxes"])
        seen.add((r["scale"], r["flip"], H < 500 or W < 500))
    assert {s for s, _, _ in seen} == {0.5, 1, 2}                # every resize branch ...
    assert {f for _, f, _ in seen} == {True, False}               # ... both flip states, and an image smaller than the crop
    assert any(small for _, _, small in seen)


def test_background_and_normalisation_constants():
    buf, boxes, paste, crop = augment.crop_image(np.full((40, 60, 3), 200, np.uint8), np.zeros((0, 4)), rng=np.random.RandomState(0))
    u8 = buf.astype(np.uint8)
    bg = u8[(np.arange(500) < paste[1])[:, None] | (np.arange(500) >= paste[3])[:, None] | (np.arange(500) < paste[0])[None] | (np.arange(500) >= paste[2])[None]]
    assert set(map(tuple, bg.reshape(-1, 3).tolist())) == {(123, 116, 103)}       # (mean * 255) truncated, processor.py:66-71
    assert (u8[paste[1]:paste[3], paste[0]:paste[2]] == 200).all()                 # int8 wrap of the paste is undone by astype(uint8)
    x = augment.to_normalized_tensor(u8)
    assert x.dtype == np.float32 and x.shape == (3, 500, 500)
    assert abs(float(x[0, 0, 0]) - (123 / 255 - 0.485) / 0.229) < 1e-6
