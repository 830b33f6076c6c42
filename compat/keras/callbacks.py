from _hdu import mod as _mod

ModelCheckpoint = _mod("keras_api").ModelCheckpoint
