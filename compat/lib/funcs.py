"""drop-in shim for lib/funcs.py (sliding-window inference)"""
from _hdu import mod as _mod

predict_tumor_inwindow = _mod("funcs").predict_tumor_inwindow
