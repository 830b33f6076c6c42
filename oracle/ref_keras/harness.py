"""oracle/ref_keras/harness.py -- TEST INFRASTRUCTURE: run the reference's OWN model constructors here.

`setup()` arranges the interpreter so that /root/reference/{denseunet,densenet,denseunet3d,hybridnet,loss}.py,
lib/custom_layers.py and the vendored Keras-2.0.8 package import UNMODIFIED under Python 3.10 without TensorFlow:

  * `keras.backend`  -> oracle/ref_keras/torch_backend.py (eager torch restatement of the TFB functions on the path)
  * `tensorflow`     -> oracle/ref_keras/fake_tf.py (the few tf.* calls of loss.py / denseunet3d.py / hybridnet.py)
  * Python-2 / old-Python names the sources use: `xrange`, `collections.Iterable` ... (aliases, nothing is edited)
  * modules denseunet.py imports but the constructors never call (medpy, skimage): empty stand-ins
  * the vendored Python-2 `yaml/` directory next to Keras is shadowed by the interpreter's own PyYAML (imported first)
  * no byte code is written into /root/reference (sys.dont_write_bytecode)

Nothing under /root/reference is copied or modified.  `build(variant, x)` returns the Keras model the reference constructor
made for input `x` (eager: its outputs are already computed); `inventory(model)` lists every layer with its class, weights,
trainable flag and the configuration values parity depends on.
"""
import argparse
import builtins
import collections
import collections.abc
import importlib
import os
import sys
import types

REF = os.environ.get("HDU_REFERENCE_ROOT", "/root/reference")
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_done = False
KB = None


def available():
    return os.path.isdir(os.path.join(REF, "Keras-2.0.8", "keras"))


def setup(floatx="float64"):
    """floatx is fixed for the life of the interpreter: the vendored engine binds `dtype=K.floatx()` as a DEFAULT ARGUMENT
    at import time (K.engine/topology.py:1375), exactly as it would under ~/.keras/keras.json"""
    global _done, KB
    if _done:
        assert KB.floatx() == floatx, "floatx was fixed at first setup()"
        return KB
    if not available():
        raise RuntimeError("reference tree not found at %s" % REF)
    sys.dont_write_bytecode = True
    for n in ("Iterable", "Mapping", "MutableMapping", "Sequence", "Callable", "Iterator", "Set", "MutableSet", "Hashable",
              "Sized", "Container"):
        if not hasattr(collections, n):
            setattr(collections, n, getattr(collections.abc, n))
    builtins.xrange = range
    import yaml  # noqa: F401  (the interpreter's PyYAML, before Keras-2.0.8/yaml (Python 2) can shadow it)
    import six  # noqa: F401
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from oracle.ref_keras import fake_tf, torch_backend
    KB = torch_backend
    KB.set_floatx(floatx)
    sys.modules["tensorflow"] = fake_tf
    # K.callbacks.py:24-26 imports the TensorBoard projector when the backend calls itself 'tensorflow'
    for name in ("tensorflow.contrib", "tensorflow.contrib.tensorboard", "tensorflow.contrib.tensorboard.plugins",
                 "tensorflow.contrib.tensorboard.plugins.projector"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["tensorflow.contrib.tensorboard.plugins"].projector = sys.modules["tensorflow.contrib.tensorboard.plugins.projector"]
    sys.modules["keras.backend"] = torch_backend
    # denseunet.py:10,24 and lib/funcs.py:3 import data-loading / post-processing packages the constructors never call
    for name in ("medpy", "medpy.io", "skimage", "skimage.transform", "skimage.measure"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["medpy.io"].load = None
    sys.modules["skimage.transform"].resize = None
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    for p in (os.path.join(REF, "lib"), REF, os.path.join(REF, "Keras-2.0.8")):
        sys.path.insert(0, p)
    import keras
    keras.backend = torch_backend
    assert os.path.realpath(keras.__file__).startswith(os.path.realpath(REF)), keras.__file__
    _done = True
    return KB


VARIANTS = ("denseunet", "densenet", "3dpart", "end2end")


def build(variant, x, learning_phase=1, floatx="float64", dropout_identity=True, seed=4321):
    """run the reference constructor of `variant` on the input tensor `x` (torch, channels-last as the reference feeds
    it: 2D (N,H,W,3); hybrids (1,H,W,D,1)).  Returns the keras Model; model.outputs[0] is the computed output."""
    K = setup(floatx)
    K.clear_session()
    K.set_learning_phase(learning_phase)
    K.DROPOUT_IDENTITY = dropout_identity
    K.seed_initializers(seed)
    K.FEED.append(x)
    if variant == "denseunet":
        mod = importlib.import_module("denseunet")            # /root/reference/denseunet.py:130 (module globals = shape)
        mod.batch_size, mod.img_deps, mod.img_rows = int(x.shape[0]), int(x.shape[1]), int(x.shape[2])
        return mod.DenseUNet(reduction=0.5)
    if variant == "densenet":
        mod = importlib.import_module("densenet")             # densenet.py:10 (takes args)
        return mod.DenseUNet(reduction=0.5, args=argparse.Namespace(b=int(x.shape[0]), input_size=int(x.shape[1])))
    args = argparse.Namespace(b=1, input_size=int(x.shape[1]), input_cols=int(x.shape[3]))
    if variant == "3dpart":
        return importlib.import_module("denseunet3d").denseunet_3d(args)        # denseunet3d.py:393
    if variant == "end2end":
        return importlib.import_module("hybridnet").dense_rnn_net(args)         # hybridnet.py:379
    raise ValueError(variant)


def loss_fn(variant):
    setup()
    mod = importlib.import_module("loss")                     # /root/reference/loss.py
    return mod.weighted_crossentropy_2ddense if variant in ("denseunet", "densenet") else mod.weighted_crossentropy


def inventory(model):
    """[(layer name, class name, trainable, config subset, [(weight name, shape, trainable)])] in model.layers order"""
    out = []
    for layer in model.layers:
        cfg = {}
        conf = layer.get_config()
        for key in ("epsilon", "momentum", "axis", "strides", "padding", "use_bias", "kernel_size", "filters", "rate", "pool_size",
                    "size", "data_format", "activation", "center", "scale"):
            if key in conf:
                v = conf[key]
                cfg[key] = list(v) if isinstance(v, tuple) else v
        tw = set(id(w) for w in layer.trainable_weights)
        ws = [(getattr(w, "_kname", None), [int(s) for s in w.shape], id(w) in tw) for w in layer.weights]
        out.append((layer.name, layer.__class__.__name__, bool(layer.trainable), cfg, ws))
    return out
