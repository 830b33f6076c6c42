"""drop-in shim for lib/funcs.py (sliding-window inference)"""
from _hdu import mod as _mod

predict_tumor_inwindow = _mod("funcs").predict_tumor_inwindow
# helpers that restate the inline post-processing of test.py:57-112 (not functions in the reference's lib/funcs.py)
liver_window_from_mask = _mod("funcs").liver_window_from_mask
segment_liver_tumor = _mod("funcs").segment_liver_tumor
