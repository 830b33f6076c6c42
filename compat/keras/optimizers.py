from _hdu import mod as _mod

SGD = _mod("keras_api").SGD
