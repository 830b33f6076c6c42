from _hdu import mod as _mod

make_parallel = _mod("keras_api").make_parallel
