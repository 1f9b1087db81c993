"""Worker of tests/test_gpu_dist.py: one data-parallel rank of the fused TrainEngine (gloo rendezvous, every rank on cuda:0 of a
1-GPU box).  Rank r trains on micro-batch r % 2 for STEPS steps; rank 0 writes the flat parameter buffer after every step."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tiny-faces-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def build(golden_path):
    from oracle.model import OracleDetectionModel, tame_init_
    from tinyfaces.models.loss import DetectionCriterion
    from tinyfaces.models.model import DetectionModel
    g = np.load(golden_path, allow_pickle=False)
    batches = [(torch.from_numpy(g[f"b{i}_img"]), torch.from_numpy(g[f"b{i}_cm"].astype(np.float32)), torch.from_numpy(g[f"b{i}_rm"]).float())
               for i in range(2)]
    m = DetectionModel(num_templates=25)
    m.load_state_dict(tame_init_(OracleDetectionModel(num_templates=25), 0).state_dict(), strict=True)
    m.set_compute_dtype(torch.float32)
    E = 25 * batches[0][1].shape[2] * batches[0][1].shape[3]
    keep = torch.ones(batches[0][0].shape[0], E, dtype=torch.uint8)
    keep[:, 128:] = 0                                     # deterministic balance sampling: the first 128 of each kind
    c = DetectionCriterion(25)
    c.inject_sampling(keep, keep)
    return m, c, batches


def main():
    golden_path, out_path, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    from tinyfaces import _hip, parallel
    from tinyfaces.engine import TrainEngine
    parallel.init_from_env("gloo")
    rank = parallel.rank()
    torch.cuda.set_device(0)
    _hip.lib().tf_set_stat_rows(0)                        # reproducible BN statistics
    m, c, batches = build(golden_path)
    eng = TrainEngine(m, c, lr=1e-4, momentum=0.9, weight_decay=5e-4, device="cuda:0", bucket_mb=10)
    assert eng._overlap is not None and len(eng._overlap["ranges"]) >= 8, "the bucketed, event-driven exchange must be active"
    img, cm, rm = [t.cuda() for t in batches[rank % 2]]
    snaps = []
    for s in range(steps):
        eng.step(img, cm.clone(), rm)
        torch.cuda.synchronize()
        snaps.append(eng.flat_p.detach().cpu().numpy().copy())
    if rank == 0:
        np.savez(out_path, *snaps, native=np.array([0 if eng._native is None else 1, eng.native_fallbacks]))
    torch.distributed.barrier()
    eng.close()


def main_native_fallback():
    """r6: `dist_worker.py native-fallback OUT`: ONE rank with a 1-rank RCCL group.  Engine A takes the native exchange (C hook) with
    TINYFACES_COMM_FAIL_BUCKET=1 in the environment: the hook refuses bucket 1 in its first step, the engine must reduce that bucket through
    torch.distributed, leave the native exchange and carry on with the ctypes callback; engine B runs the same steps on the torch.distributed
    exchange from the start.  Both must end with the same parameters (the sum over one rank is the identity)."""
    out_path = sys.argv[2]
    from tinyfaces import _hip, parallel
    from tinyfaces.engine import TrainEngine
    golden_path = os.path.join(ROOT, "tests", "golden", "trainer.npz")
    assert parallel.init_from_env("nccl"), "TINYFACES_FORCE_DIST=1 must be set: a 1-rank RCCL group"
    torch.cuda.set_device(0)
    _hip.lib().tf_set_stat_rows(0)
    res = {}
    for tag, native in (("a", True), ("b", False)):
        m, c, batches = build(golden_path)
        eng = TrainEngine(m, c, lr=1e-4, momentum=0.9, weight_decay=5e-4, device="cuda:0", bucket_mb=10, native_exchange=native)
        res[tag + "_native_at_start"] = 0 if eng._native is None else 1
        img, cm, rm = [t.cuda() for t in batches[0]]
        for s in range(3):
            eng.step(img, cm.clone(), rm)
            if s == 0:
                res[tag + "_native_after_step1"] = 0 if eng._native is None else 1
        torch.cuda.synchronize()
        res[tag + "_flat"] = eng.flat_p.detach().cpu().numpy().copy()
        res[tag + "_fallbacks"] = eng.native_fallbacks
        eng.close()
    np.savez(out_path, **res)
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main_native_fallback() if sys.argv[1] == "native-fallback" else main()
