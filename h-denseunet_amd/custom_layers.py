"""Drop-in for lib/custom_layers.py: `Scale` (out = in*gamma + beta per channel, lib/custom_layers.py:10-74).
In this framework Scale never runs as its own kernel: it is folded with the preceding BatchNormalization into one
per-channel affine applied inside the consumer conv's operand gather (engine.BNLayer).  The class keeps the
reference's constructor signature and records the two Keras weights [gamma, beta]."""
import numpy as np


class Scale:
    def __init__(self, weights=None, axis=-1, momentum=0.9, beta_init="zero", gamma_init="one", **kwargs):
        self.axis, self.momentum = axis, momentum
        self.beta_init, self.gamma_init = beta_init, gamma_init
        self.initial_weights = weights
        self.name = kwargs.get("name")
        self.trainable = kwargs.get("trainable", True)

    def build(self, input_shape):
        c = int(input_shape[self.axis])
        self.gamma = np.ones(c, np.float32)
        self.beta = np.zeros(c, np.float32)
        if self.initial_weights is not None:
            self.gamma, self.beta = [np.asarray(w, np.float32) for w in self.initial_weights]

    def call(self, x):
        shape = [1] * x.ndim
        shape[self.axis] = -1
        return x * self.gamma.reshape(shape) + self.beta.reshape(shape)

    def get_config(self):
        return {"momentum": self.momentum, "axis": self.axis}
