"""CPU oracle for the tiny-faces hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy / torch-CPU restatement of the reference's algorithm
(varunagrawal/tiny-faces-pytorch) for the path named in BASELINE.json `north_star`.
Every function cites the reference file:line it follows.

Rules (enforced by tests/test_no_oracle_in_product.py):
  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
    import anything from `oracle/`;
  * nothing under `tiny-faces-pytorch_amd/` imports it -- the product path is the HIP
    library and fails loudly when that is missing.

Pinning status (see DESIGN.md "Oracle"):
  * reference-owned arithmetic (dense_overlap, get_heatmaps/get_regression/get_padding,
    get_bboxes/regression_refinement, balance_sampling, DetectionCriterion, the head /
    crop logic of DetectionModel.forward, get_detections control flow) is PINNED: the
    golden vectors in tests/golden/ were produced by importing the reference's own
    source in the build container (oracle/tools/make_golden.py) and the restatements
    here are asserted equal to them.
  * the training augmentation in front of the path (oracle/augment.py: WIDERFace.process_inputs, DataProcessor.crop_image)
    is PINNED the same way (tests/golden/augment.npz), and the PIL BILINEAR resize underneath it -- third-party Pillow 12.2,
    which IS installed -- is asserted bit-equal to Pillow itself (tests/test_oracle_augment.py).
  * torchvision-owned arithmetic (resnet101 trunk, ops.nms, transforms) is a
    restatement of third-party torchvision 0.18 whose source is not in /root/reference
    and which is not installed here: PARITY UNPINNED at that boundary.  The trunk is
    built only from torch.nn ops (which ARE installed), and nms follows the published
    torchvision CPU kernel semantics.
"""
