"""r5 diagnostic, part 4: ONE level, a fresh constant_weights session per forward (weights re-packed + BN re-folded every time) against forwards inside one session."""
import os, sys, numpy as np, torch
ROOT = os.environ.get("DIAG_ROOT") or os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, ROOT + "/tiny-faces-pytorch_amd", ROOT + "/tests"]
from tinyfaces.models.model import DetectionModel
from oracle.model import OracleDetectionModel, tame_init_
m = DetectionModel(num_templates=25)
m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
m = m.cuda().eval().set_compute_dtype(torch.bfloat16)
gen = torch.Generator().manual_seed(0)
R = int(os.environ.get("REPS", "40"))
for hw in ((1400, 1866), (700, 933), (500, 500)):
    x = torch.randn(1, 3, *hw, generator=gen).cuda()
    with torch.no_grad():
        with m.constant_weights(reserve=(1, *hw)):
            ref = m(x).clone()
        torch.cuda.synchronize()
        for mode in ("new session per forward", "no session (packs every forward)", "one session"):
            bad = 0; info = []
            if mode == "one session":
                with m.constant_weights(reserve=(1, *hw)):
                    for _ in range(R):
                        y = m(x)
                        if not torch.equal(y, ref): bad += 1
            else:
                for _ in range(R):
                    if mode.startswith("new"):
                        with m.constant_weights(reserve=(1, *hw)):
                            y = m(x)
                    else:
                        y = m(x)
                    if not torch.equal(y, ref):
                        bad += 1
                        d = (y - ref).abs()[0].amax(0); pos = torch.nonzero(d > 0)
                        if len(info) < 3: info.append((int(pos.shape[0]), pos[0].tolist(), float(d.max())))
            print(hw, mode, "-> mismatching forwards", bad, "of", R, info, flush=True)
