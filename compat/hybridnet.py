"""drop-in shim with the reference's module name (hybridnet.py): re-exports the MI355X implementation"""
from _hdu import mod as _mod

_m = _mod("hybridnet")
globals().update({k: v for k, v in vars(_m).items() if not k.startswith("_")})
