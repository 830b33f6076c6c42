"""minimal `keras` namespace holding exactly what the reference's training scripts import from Keras for the hot
path (train_2ddense.py:13-19, train_hybrid.py:13-21): SGD, ModelCheckpoint, make_parallel, backend.set_image_dim_ordering."""
