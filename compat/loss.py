"""drop-in shim with the reference's module name (loss.py): re-exports the MI355X implementation"""
from _hdu import mod as _mod

_m = _mod("loss")
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("_")})
