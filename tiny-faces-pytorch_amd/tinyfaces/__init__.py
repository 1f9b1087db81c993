"""tinyfaces -- MI355X-native (gfx950) implementation of the Tiny Faces detection hot path behind
the call surface of varunagrawal/tiny-faces-pytorch (import paths tinyfaces.models.model,
tinyfaces.models.loss, tinyfaces.trainer, tinyfaces.evaluation, tinyfaces.datasets).

Put `tiny-faces-pytorch_amd/` on PYTHONPATH and the reference's main.py / evaluate_model.py
resolve their `tinyfaces.*` imports here.  The compute path is libtinyfaces_hip.so only."""
__version__ = "0.1.0"
