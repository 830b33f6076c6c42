"""Drop-in for lib/custom_layers.py: `Scale` (out = in*gamma + beta per channel, lib/custom_layers.py:10-74).

A `Scale` never runs as a kernel of its own here: the network builders (models.py) create one per reference Scale layer and
hand it to the BatchNormalization it follows (engine.BNLayer(scale=...)), which owns the two Keras weights [gamma, beta] under
the Scale's layer name and folds them with its own into ONE per-channel affine a = sg*g*rstd, b = sg*(beta - mean*g*rstd) + sb
applied by the consumer (conv operand path / materialise pass); the backward produces d(gamma), d(beta) of the Scale from the
same two sums as the BatchNormalization's (csrc/rowops.hip: bn_coef_channel).  This class is that descriptor: the reference's
constructor signature, the layer name and the `trainable` flag (frozen in `denseunet_3d`, trainable elsewhere).
`call` is the plain numpy statement of the layer, used by the tests as the definition of what the folded affine must equal."""
import numpy as np


class Scale:
    def __init__(self, weights=None, axis=-1, momentum=0.9, beta_init="zero", gamma_init="one", **kwargs):
        self.axis, self.momentum = axis, momentum
        self.beta_init, self.gamma_init = beta_init, gamma_init
        self.initial_weights = weights
        self.name = kwargs.get("name")
        self.trainable = kwargs.get("trainable", True)
        self.gamma = self.beta = None        # engine.Param objects once a BNLayer has adopted the layer

    def build(self, input_shape):
        c = int(input_shape[self.axis])
        self.gamma = np.ones(c, np.float32)
        self.beta = np.zeros(c, np.float32)
        if self.initial_weights is not None:
            self.gamma, self.beta = [np.asarray(w, np.float32) for w in self.initial_weights]

    def call(self, x):
        shape = [1] * x.ndim
        shape[self.axis] = -1
        return x * np.asarray(self.gamma).reshape(shape) + np.asarray(self.beta).reshape(shape)

    def get_config(self):
        return {"momentum": self.momentum, "axis": self.axis}
