"""Finds the planted head of tests/planted.py and writes tests/golden/planted_head.json.  Test infrastructure (runs on the CPU, uses the oracle).

For each of the eight admissible templates: a res3 channel of the tame-init oracle trunk whose values over the three pyramid levels of the planted
image have a gap of >= 2 * K * e below their 1..6 largest, e = that channel's largest bf16 deviation (ESTIMATED here by running the oracle trunk in
torch bfloat16 on the CPU; the GPU test measures the real one and asserts its own margins).  tau = the middle of the gap.  A row is kept only if the
whole candidate list so far stays robust (tests/planted.py:robustness) against a logit error of K * EPS (each row's gain = EPS / e).

    python oracle/tools/make_planted_head.py
"""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd"), os.path.join(ROOT, "tests")]

from oracle.model import OracleDetectionModel, tame_init_      # noqa: E402
from oracle.nms import nms as onms                                  # noqa: E402
from oracle.refstub import Compose, Normalize, ToTensor, resize, to_pil_image   # noqa: E402
from planted import VALID_TEMPLATES, planted_image, robustness      # noqa: E402
from tinyfaces.datasets.templates import load_templates             # noqa: E402
from tinyfaces.ops import RF                                        # noqa: E402

K = float(os.environ.get('PLANT_K', 8.0))
SCALES = (0.5, 1.0, 2.0)


def res3(model, x):
    m = model.model
    with torch.no_grad():
        x = m.maxpool(m.relu(m.bn1(m.conv1(x))))
        return m.layer2(m.layer1(x))


def boxes_of(t, fy, fx, scale, templates):
    cy, cx = fy * RF["stride"][0] + RF["offset"][0], fx * RF["stride"][1] + RF["offset"][1]
    cw, ch = templates[t, 2] - templates[t, 0] + 1, templates[t, 3] - templates[t, 1] + 1
    return np.stack([cx - cw / 2, cy - ch / 2, cx + cw / 2, cy + ch / 2], 1) / scale


def main():
    torch.set_num_threads(os.cpu_count() or 8)
    templates = load_templates()
    om = tame_init_(OracleDetectionModel(num_templates=25), 0).eval()
    ob = copy.deepcopy(om).to(torch.bfloat16)
    u8 = planted_image(decay=float(os.environ.get('PLANT_DECAY', 0.93)), k=int(os.environ.get('PLANT_N', 12)), seed=int(os.environ.get('PLANT_SEED', 0)))
    img = torch.from_numpy(u8).permute(2, 0, 1).float().div(255)
    tf = Compose([ToTensor(), Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    image = to_pil_image(img)
    f, e = [], []
    for s in SCALES:
        x = tf(resize(image, int(min(image.size) * s))).unsqueeze(0).float()
        a = res3(om, x)[0].numpy()
        b = res3(ob, x.to(torch.bfloat16))[0].float().numpy()
        f.append(a)
        e.append(np.abs(a - b).reshape(a.shape[0], -1).max(axis=1))
        print(f"level x{s}: res3 {a.shape}, max {a.max():.3f}, bf16 deviation (CPU estimate) {e[-1].max():.4f}")
    e = np.max(e, axis=0)                                           # per channel, over the levels
    C = f[0].shape[0]
    EPS = 0.02                                                      # every row's logit carries the same estimated error: gain = EPS / e[channel]
    pooled = [np.sort(np.concatenate([a[c].ravel() for a in f]))[::-1][:64] for c in range(C)]
    order = sorted(range(C), key=lambda c: -(pooled[c][0] / max(e[c], 1e-9)))
    rows, used = [], set()

    def candidates(rows, floor):
        """Every position with logit > floor, in the decode's order (level, y, x, template)."""
        out = []
        for li, s in enumerate(SCALES):
            lv = []
            for t, c, tau, gain in rows:
                lg = gain * (f[li][c] - tau)
                fy, fx = np.nonzero(lg > floor)
                lv.append(np.concatenate([boxes_of(t, fy, fx, s, templates), lg[fy, fx][:, None], fy[:, None], fx[:, None], np.full((fy.size, 1), t)], 1))
            lv = np.concatenate(lv)
            out.append(lv[np.lexsort((lv[:, 7], lv[:, 6], lv[:, 5]))])
        return np.concatenate(out)

    for t in VALID_TEMPLATES:
        options = []
        for c in range(C):
            if c in used or e[c] == 0:
                continue
            v = pooled[c]
            gaps = v[:12] - v[1:13]
            ok = [j for j in range(12) if gaps[j] >= 2 * K * e[c]]
            if ok:
                options.append((max(ok), gaps[max(ok)] / e[c], c))
        fails = []
        for j, _, c in sorted(options, reverse=True):
            v = pooled[c]
            tau = float(np.float32((v[j] + v[j + 1]) / 2))
            trial = rows + [[t, c, tau, float(np.float32(EPS / e[c]))]]
            M = K * EPS
            allc = candidates(trial, -4 * M)
            cand = allc[allc[:, 4] > 0]
            keep = onms(cand[:, :4], cand[:, 4], 0.3)
            bad = robustness(cand, allc, keep, 0.0, M)
            if bad == (0, 0, 0, 0):
                rows, used = trial, used | {c}
                print(f"template {t}: channel {c}, tau {tau:.5f}, {j + 1} candidates, gap {(v[j] - v[j + 1]) / e[c]:.0f} e; total {cand.shape[0]} candidates, {len(keep)} survivors, M {M:.3f}")
                break
            fails.append((bad, c, j))
        else:
            print(f"template {t}: no channel found; first failures {fails[:6]} of {len(fails)}")
    plan = {"rows": rows, "note": "oracle/tools/make_planted_head.py; rows of (template, channel, tau, gain): logit[template] = gain * (res3[channel] - tau); bf16 deviation estimated on the CPU"}
    with open(os.path.join(ROOT, "tests", "golden", "planted_head.json"), "w") as fh:
        json.dump(plan, fh, indent=1)
    print("wrote", len(rows), "rows")


if __name__ == "__main__":
    main()
