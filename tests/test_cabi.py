"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/tinyfaces_hip.h
declares (no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import importlib.util
    spec = importlib.util.spec_from_file_location("tf_build", os.path.join(ROOT, "tiny-faces-pytorch_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(verbose=False)


def _declared():
    src = open(os.path.join(ROOT, "include", "tinyfaces_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built, hip):
    import ctypes
    l = ctypes.CDLL(built)
    declared = _declared()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(l, name), f"{name} declared in include/tinyfaces_hip.h but not exported"
    listed = hip.symbols()
    assert set(declared) == set(listed)
    assert hip.lib().tf_version() >= 100


def test_binding_signatures_cover_the_abi(hip):
    assert set(hip._SIGNATURES) == set(hip.symbols())


def test_executor_parameter_table_matches_state_dict(hip):
    from tinyfaces.models.model import DetectionModel
    m = DetectionModel(num_templates=25)
    sd = m.state_dict()
    l = hip.lib()
    n = l.tf_detnet_num_params()
    names = [l.tf_detnet_param_name(i).decode() for i in range(n)]
    assert len(set(names)) == n == 475
    for i, k in enumerate(names):
        assert k in sd, k
        assert sd[k].numel() == l.tf_detnet_param_numel(i, 125), k
    # everything in the state_dict that the executor does not read is dead weight of the reference (D4) or bookkeeping
    rest = [k for k in sd if k not in names]
    assert all(k.startswith("model.fc.") or k.endswith("num_batches_tracked") for k in rest)


def test_workspace_and_shape_queries(hip):
    import ctypes as C
    l = hip.lib()
    h3, w3 = C.c_int(), C.c_int()
    for (h, w), exp in {(500, 500): (63, 63), (501, 503): (63, 63), (480, 640): (60, 80), (960, 1280): (120, 160),
                        (64, 64): (8, 8), (1000, 1000): (125, 125)}.items():      # SURVEY.md section 3.3 [probe]
        l.tf_detnet_out_shape(h, w, C.byref(h3), C.byref(w3))
        assert (h3.value, w3.value) == exp
    train = l.tf_detnet_workspace_bytes(hip.TF_BF16, 12, 500, 500, 125, 1)
    evalb = l.tf_detnet_workspace_bytes(hip.TF_BF16, 12, 500, 500, 125, 0)
    assert 1e9 < evalb < train < 2e10
    assert l.tf_nms_workspace_bytes(4096) > 4096 * 64 * 8
    assert l.tf_criterion_workspace_bytes(12, 25, 63, 63) > 0 and l.tf_targets_workspace_bytes(100) >= 1200
