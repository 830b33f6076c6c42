"""The slice of the keras.models.Model surface the reference scripts use, over the engine.

  compile(optimizer=SGD(...), loss=[fn])              K.engine/training.py:570
  train_on_batch(x, y) / fit_generator(...)           K.engine/training.py:1715 / :1831
  predict(x, batch_size, verbose)                     K.engine/training.py:1659
  load_weights / save_weights / save                  K.engine/topology.py:2590 / :2555 (npz container: h5py is
                                                      not available in this image; same layer names and per-layer
                                                      weight order as Keras' HDF5 groups)
  get_weights / set_weights, .inputs/.outputs shapes, summary()
"""
import os
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .engine import Ctx, LossLayer
from .lib import HDU_BF16, HDU_F32
from . import models as _m


def _graph_capture(graph):
    """torch.cuda.graph(...) for one captured step.  Under an initialised NCCL (= RCCL) process group torch's watchdog thread polls
    events while this thread captures, which aborts a capture in the default "global" error mode: only
    THEN is the capture thread-local.  A single-GPU capture keeps the runtime's global protection against capture-unsafe calls from
    other threads.  torch versions without the keyword take the plain form."""
    mode = "global"
    dist = torch.distributed
    if dist.is_available() and dist.is_initialized() and str(dist.get_backend()).lower().find("nccl") >= 0:
        mode = "thread_local"
    try:
        return torch.cuda.graph(graph, capture_error_mode=mode)
    except TypeError:
        return torch.cuda.graph(graph)


class SGD:
    """K.optimizers.py:137-186 (Nesterov momentum form; `decay` as in :160-163)."""

    def __init__(self, lr=0.01, momentum=0.0, decay=0.0, nesterov=False, **kw):
        if not nesterov:
            raise NotImplementedError("the reference trains with SGD(nesterov=True) (train_2ddense.py:181); only that form is built")
        self.lr, self.momentum, self.decay, self.nesterov = float(lr), float(momentum), float(decay), True
        self.iterations = 0


def weighted_crossentropy(y_true, y_pred):          # loss.py:5 -- marker object: the kernel computes it
    raise RuntimeError("marker function: pass it to Model.compile(loss=[...]); the HIP loss kernel computes it")


def weighted_crossentropy_2ddense(y_true, y_pred):  # loss.py:27
    raise RuntimeError("marker function: pass it to Model.compile(loss=[...]); the HIP loss kernel computes it")


def _is_io_rank():
    """one process per GPU under data parallelism / depth sharding: checkpoints, the loss history and progress lines
    are written by rank 0 only (the reference's single-process make_parallel wrote each file once)"""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


class History:
    def __init__(self):
        self.history = {"loss": []}


class Model:
    def __init__(self, kind, args_b, input_size, input_cols=None, dtype="bf16", variant=None, name=None,
                 nb_layers2d=(6, 12, 36, 24), nb_layers3d=(3, 4, 12, 8), seed=4321, shard=None):
        self.kind = kind                  # "2d" | "hybrid" | "3d"
        self.name = name
        self.dtype = HDU_BF16 if dtype in ("bf16", HDU_BF16) else HDU_F32
        self.variant = variant
        ctx = self.ctx = Ctx(self.dtype, None)
        dt = self.dtype
        if kind == "2d":
            N, H, W = args_b, input_size, input_size
            self.input_shape = (N, H, W, 3)
            self.x_stage = torch.zeros(N * H * W * 3, dtype=torch.float32, device=ctx.dev)
            self.x_in = ctx.new_var(N, 1, H, W, ops.cpad(3, dt))
            ctx.fwd.append(lambda: ops.cast_pad(self.x_stage, N * H * W, 3, self.x_in.act))
            r = _m.build_dense_unet_2d(ctx, self.x_in, variant=variant, nb_layers=nb_layers2d)
            self.logits = r["logits"]
            self.loss_layer = LossLayer(ctx, self.logits, [(0, N * H * W)])
            self.output_shape = (N, H, W, 3)
        elif kind == "hybrid":
            if args_b != 1:
                raise ValueError("the hybrid nets index the batch axis as the slice index: b must be 1 "
                                 "(hybridnet.py:359-364,400-406)")
            H = W = input_size
            D = input_cols
            ctx.shard = shard
            sharded = shard is not None and shard.world > 1
            self.input_shape = (1, H, W, D, 1)
            if sharded:
                # input_cols = depth planes held by THIS rank; one raw CT plane of each depth neighbour rides along for
                # the 2.5D slabs (denseunet3d.py:399-409): vol_h = [halo][D planes][halo], vol = its interior
                self.vol_h = torch.zeros((D + 2) * H * W, dtype=torch.float32, device=ctx.dev)
                self.vol = self.vol_h[H * W:(D + 1) * H * W]
                self.logits = _m.build_hybrid(ctx, self.vol_h, D, H, W, variant=variant, nb_layers2d=nb_layers2d,
                                              nb_layers3d=nb_layers3d)
            else:
                self.vol = torch.zeros(D * H * W, dtype=torch.float32, device=ctx.dev)
                self.logits = _m.build_hybrid(ctx, self.vol, D, H, W, variant=variant, nb_layers2d=nb_layers2d,
                                              nb_layers3d=nb_layers3d)
            # loss.py:6-7 hard-codes depth slices 1:7 (of the WHOLE volume: a shard takes its part of that range)
            g0 = shard.rank * D if sharded else 0
            world = shard.world if sharded else 1
            lo, hi = max(1, g0), min(7, D * world, g0 + D)
            ranges = [((lo - g0) * H * W, (hi - lo) * H * W)] if hi > lo else []
            self.loss_layer = LossLayer(ctx, self.logits, ranges)
            if sharded:   # normalised by the global voxel count: count * world == (min(7, D_global) - 1) * H * W
                self.loss_layer.count = (min(7, D * world) - 1) * H * W / float(world)
            self.output_shape = (1, H, W, D, 3)
        elif kind == "3d":
            # stand-alone DenseNet3D on a 4-channel volume; with `shard` this rank holds input_cols LOCAL depth planes
            if args_b != 1:
                raise ValueError("one volume per step")
            H = W = input_size
            D = input_cols
            ctx.shard = shard
            self.input_shape = (1, H, W, D, 4)
            self.x_stage = torch.zeros(D * H * W * 4, dtype=torch.float32, device=ctx.dev)
            self.x_in = ctx.new_var(1, D, H, W, ops.cpad(4, dt))
            ctx.fwd.append(lambda: ops.cast_pad(self.x_stage, D * H * W, 4, self.x_in.act))
            self.logits = _m.build_dense_net_3d_standalone(ctx, self.x_in, nb_layers=nb_layers3d)
            self.loss_layer = LossLayer(ctx, self.logits, [(0, D * H * W)])
            self.output_shape = (1, H, W, D, 3)
        else:
            raise ValueError(kind)
        self.logits.require_grad()
        ctx.finalize(seed)
        self.out_stage = torch.zeros(self.logits.act.M * 3, dtype=torch.float32, device=ctx.dev)
        self.optimizer = None
        self.world_size = 1
        self._graph = None
        self._graph_hparams = None
        self._eager_steps = 0            # eager training steps run so far (capture_graph needs two before it captures)
        self._allreduce = None
        self._allreduce_async = None
        self._buckets = None
        self.inputs = [self.input_shape]
        self.outputs = [self.output_shape]
        self.stop_training = False
        self._pending_iterations = None
        self._loss_names = None

    # ------------------------------------------------------------------ boundary layout conversion
    def _upload_x(self, x):
        x = torch.as_tensor(np.asarray(x, dtype=np.float32)) if not torch.is_tensor(x) else x.float()
        if tuple(x.shape) != tuple(self.input_shape):
            raise ValueError("expected input of shape %s, got %s" % (self.input_shape, tuple(x.shape)))
        if self.kind == "2d":
            self.x_stage.copy_(x.reshape(-1).to(self.ctx.dev), non_blocking=True)
        elif self.kind == "3d":  # (1,H,W,D,4) -> depth-major [D][H][W][4]
            self.x_stage.copy_(x[0].permute(2, 0, 1, 3).contiguous().reshape(-1).to(self.ctx.dev))
        else:  # (1,H,W,D,1) -> depth-major [D][H][W]
            self.vol.copy_(x[0, :, :, :, 0].permute(2, 0, 1).contiguous().reshape(-1).to(self.ctx.dev))
            if getattr(self, "vol_h", None) is not None:
                from . import shard as _sh
                H, D = self.input_shape[1], self.input_shape[3]
                _sh.exchange_ct_planes(self.ctx.shard, self.vol_h, D, H * H)

    def _labels_internal(self, y):
        y = np.asarray(y)
        if self.kind == "2d":
            want = self.input_shape[:3] + (1,)
            if tuple(y.shape) != want:
                raise ValueError("expected labels of shape %s, got %s" % (want, tuple(y.shape)))
            return y.reshape(-1)
        want = self.input_shape[:4] + (1,)
        if tuple(y.shape) != want:
            raise ValueError("expected labels of shape %s, got %s" % (want, tuple(y.shape)))
        return np.ascontiguousarray(y[0, :, :, :, 0].transpose(2, 0, 1)).reshape(-1)

    def _download_logits(self):
        a = self.logits.act
        ops.cast_out(a, 3, self.out_stage)
        o = self.out_stage.reshape(a.N, a.D, a.H, a.W, 3)
        if self.kind == "2d":
            return o[:, 0]
        return o.permute(0, 2, 3, 1, 4)   # (1,D,H,W,3) -> (1,H,W,D,3)

    # ------------------------------------------------------------------ Keras surface
    def compile(self, optimizer, loss=None, **kw):
        if not isinstance(optimizer, SGD):
            raise TypeError("optimizer must be SGD (the reference uses SGD(lr=1e-3, momentum=0.9, nesterov=True))")
        self.optimizer = optimizer
        if getattr(self, "_pending_iterations", None) is not None:      # optimizer state was loaded before compile()
            optimizer.iterations = self._pending_iterations
            self._pending_iterations = None
        fns = loss if isinstance(loss, (list, tuple)) else [loss]
        self._loss_names = [getattr(fn, "__name__", str(fn)) for fn in fns if fn is not None]
        for fn in fns:
            if fn is not None and getattr(fn, "__name__", "") not in ("weighted_crossentropy", "weighted_crossentropy_2ddense"):
                raise ValueError("loss must be loss.weighted_crossentropy / weighted_crossentropy_2ddense")

    def set_data_parallel(self, world_size, allreduce, allreduce_async=None, bucket_fractions=(0.85,)):
        """one process per GPU: `allreduce(flat_grad_tensor)` sums gradients over ranks (RCCL); the loss of the
        merged batch is a mean over ALL towers (multi_gpu.py:65-68 + loss.py:44).
        With `allreduce_async(t) -> work` (work.wait()) the exchange is bucketed: the backward pass is cut where the
        first-completed `bucket_fractions` of the parameters (the decoder and the late dense blocks: most of the
        bytes) have their final gradient; that bucket's all-reduce runs on the communication stream while the early
        layers' backward still computes, and only the small last bucket is exposed."""
        self.world_size = world_size
        self._allreduce = allreduce
        self._allreduce_async = allreduce_async
        self._buckets = self.ctx.grad_buckets(bucket_fractions) if allreduce_async is not None else None
        if self._buckets is not None:
            self.ctx.set_batch_wgrad(False)   # a bucket's gradients must be final when its backward segment ends
        self.loss_layer.global_scale = 1.0 / world_size

    def _step_head(self):
        ctx = self.ctx
        ctx.learning_phase = 1
        ctx.step_zero()          # accumulators, flat gradient, dropout seed: one launch, no torch kernels in the step
        ctx.prep_weights()
        ctx.run_forward()
        self.loss_layer.run(True)

    def _step_device(self):
        self._step_head()
        self.ctx.run_backward()

    def _exchange_bucket(self, bk):
        lo, hi = bk[2], bk[3]
        return self._allreduce_async(self.ctx.G[lo:hi]) if hi > lo else None

    def _step_update(self):
        ctx, opt = self.ctx, self.optimizer
        lr = opt.lr
        if opt.decay > 0:
            lr = lr * (1.0 / (1.0 + opt.decay * opt.iterations))
        ops.sgd_nesterov(ctx.P[:ctx.n_trainable], ctx.V[:ctx.n_trainable], ctx.G[:ctx.n_trainable], lr, opt.momentum)
        opt.iterations += 1

    def train_step_resident(self):
        """one fwd+bwd+SGD step on the inputs / labels already resident in HBM (what bench.py times)."""
        if self.optimizer is None:
            raise RuntimeError("compile() the model first")
        if self._graph is not None and self._graph_hparams != (self.optimizer.lr, self.optimizer.momentum):
            # the captured SGD launch carries lr / momentum as kernel arguments: an LR-schedule callback that assigns
            # optimizer.lr would otherwise be ignored silently.  All buffers are static, so re-capturing needs no warm-up.
            self.capture_graph(warmup=0)
        if self._graph is not None:
            g_fb, g_upd = self._graph
            if isinstance(g_fb, list):               # bucketed data parallel: one graph per backward segment
                works = []
                for g, bk in zip(g_fb, self._buckets):
                    g.replay()
                    works.append(self._exchange_bucket(bk))
                for w in works:
                    if w is not None:
                        w.wait()
            else:
                g_fb.replay()
                if self._allreduce is not None:
                    self._allreduce(self.ctx.G[:self.ctx.n_trainable])
            if g_upd is not None:
                g_upd.replay()
            self.optimizer.iterations += 1
            return
        if self._buckets is not None:
            self._step_head()
            works = []
            for bk in self._buckets:
                self.ctx.run_backward((bk[0], bk[1]))
                works.append(self._exchange_bucket(bk))
            for w in works:
                if w is not None:
                    w.wait()
        else:
            self._step_device()
            if self._allreduce is not None:
                self._allreduce(self.ctx.G[:self.ctx.n_trainable])
        self._step_update()
        self._eager_steps += 1

    # Capture mode: "thread_local".  In the default "global" mode ANY thread's capture-unsafe runtime call fails while this thread
    # captures -- and under data parallelism torch's ProcessGroupNCCL watchdog thread polls hipEventQuery on the works of the
    # parameter broadcast / the warm-up all-reduces: when a poll lands inside the capture the process aborts with
    # hipErrorStreamCaptureUnsupported (round 5: seen on 2 of 3 runs of the world-1 RCCL bench path once the step had become
    # shorter; it would have cost the driver's multi-GPU line).  The capturing thread's own calls are still checked.
    def capture_graph(self, warmup=2):
        """capture the step into hipGraphs.  Single GPU: ONE graph (fwd + bwd + SGD).  Data parallel: the gradient
        all-reduce stays an eager RCCL call between two graphs (fwd+bwd | SGD update) -- no collective is captured."""
        if self.optimizer is None:
            raise RuntimeError("compile() the model first")
        if self.optimizer.decay > 0:
            raise RuntimeError("graph capture freezes the learning rate; decay>0 is not supported with it")
        # The step's device tables (ops.ZeroPlan of the step head, ops.BnBwdPlan of the deferred finalizes) are built lazily
        # by the first TWO eager steps (the first backward registers the partially written gradient slabs, the second step
        # head builds the table with them) with synchronous host-to-device copies: that must not happen inside the capture.
        warmup = max(warmup, 2 - self._eager_steps)
        self._graph = None                 # (a re-capture: the warm-up below must run eagerly)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.train_step_resident()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        it0 = self.optimizer.iterations
        g_fb = torch.cuda.CUDAGraph()
        g_upd = None
        if self._allreduce is None:
            with _graph_capture(g_fb):
                self._step_device()
                self._step_update()
        else:
            if self._buckets is not None:
                g_fb = []
                for i, bk in enumerate(self._buckets):
                    g = torch.cuda.CUDAGraph()
                    with _graph_capture(g):
                        if i == 0:
                            self._step_head()
                        self.ctx.run_backward((bk[0], bk[1]))
                    g_fb.append(g)
            else:
                with _graph_capture(g_fb):
                    self._step_device()
            g_upd = torch.cuda.CUDAGraph()
            with _graph_capture(g_upd):
                self._step_update()
        self.optimizer.iterations = it0
        self._graph = (g_fb, g_upd)
        self._graph_hparams = (self.optimizer.lr, self.optimizer.momentum)

    def train_on_batch(self, x, y=None, **kw):
        if hasattr(x, "fill_model"):
            # augment.DeviceBatch: the sample assembly kernels write input and labels straight into the model's buffers
            x.fill_model(self)
        else:
            self._upload_x(x)
            self.loss_layer.set_labels(self._labels_internal(y))
        self.train_step_resident()
        return self.loss_value()

    def loss_value(self):
        """loss of the last step (mean over the global batch under data parallelism)"""
        v = self.loss_layer.loss_sum.clone()
        if self._allreduce is not None:
            self._allreduce(v)
        return float(v.item()) / float(self.loss_layer.count * self.world_size)

    def predict(self, x, batch_size=None, verbose=0):
        """forward with learning_phase=0: moving BN statistics everywhere, dropout off."""
        ctx = self.ctx
        self._upload_x(x)
        ctx.learning_phase = 0
        try:
            ctx.prep_weights()
            ctx.run_forward()
        finally:
            ctx.learning_phase = 1
        return self._download_logits().cpu().numpy()

    def forward_train_mode(self, x):
        """logits of the training-phase forward (batch statistics), without touching weights; moving statistics
        ARE updated exactly as in a training step.  Used by the parity tests."""
        self._upload_x(x)
        self.ctx.learning_phase = 1
        self.ctx.prep_weights()
        self.ctx.run_forward()
        return self._download_logits().cpu().numpy()

    def __call__(self, x):
        return self.predict(x)

    def fit_generator(self, generator, steps_per_epoch, epochs=1, verbose=1, callbacks=None, max_queue_size=10,
                      workers=1, use_multiprocessing=False, **kw):
        hist = History()
        callbacks = callbacks or []
        for cb in callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
        for epoch in range(epochs):
            losses = []
            for _ in range(int(steps_per_epoch)):
                x, y = next(generator)
                losses.append(self.train_on_batch(x, y))
            logs = {"loss": float(np.mean(losses))}
            hist.history["loss"].append(logs["loss"])
            if verbose and _is_io_rank():
                print("Epoch %d/%d - loss: %.4f" % (epoch + 1, epochs, logs["loss"]))
                # the author's patch to ProgbarLogger.on_epoch_end (K.callbacks.py:28,311-314): the epoch loss is
                # appended to ./Experiments/history/lossepoch.txt (the scripts create that directory,
                # train_2ddense.py:190-200); skipped when it does not exist instead of raising
                hist_dir = os.path.join("Experiments", "history")
                if os.path.isdir(hist_dir):
                    with open(os.path.join(hist_dir, "lossepoch.txt"), "a") as f:
                        f.write("%.4f\n" % logs["loss"])
            for cb in callbacks:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        return hist

    # ------------------------------------------------------------------ weights
    def layer_names(self):
        return list(self.ctx.by_layer.keys())

    def get_weights_dict(self):
        return OrderedDict((n, self.ctx.get_layer_weights(n)) for n in self.ctx.by_layer)

    def set_weights_dict(self, d, strict=True):
        for n, arrs in d.items():
            if n not in self.ctx.by_layer:
                if strict:
                    raise ValueError("no layer named %r in model %s" % (n, self.name))
                continue
            self.ctx.set_layer_weights(n, arrs)
        self.ctx.unprime_stats()
        if self._graph is not None:
            self._graph = None      # a captured step contains the primed (epilogue-statistics) launch list

    def get_grads_dict(self):
        return OrderedDict((n, self.ctx.get_layer_grads(n)) for n in self.ctx.by_layer)

    # Keras 2.0.8 weight names inside a layer's HDF5 group (K.layers/convolutional.py:128-143, normalization.py:97-123,
    # lib/custom_layers.py:53-57: the Scale layer names its variables '<layer>_gamma' / '<layer>_beta')
    def _weight_names(self, layer):
        kind = self.ctx.layer_kind[layer]
        n = len(self.ctx.by_layer[layer])
        if kind == "conv":
            return ["%s/kernel:0" % layer, "%s/bias:0" % layer][:n]
        if kind == "bn":
            return ["%s/%s:0" % (layer, w) for w in ("gamma", "beta", "moving_mean", "moving_variance")]
        if kind == "scale":
            return ["%s/%s_gamma:0" % (layer, layer), "%s/%s_beta:0" % (layer, layer)]
        raise ValueError(kind)

    def _keras_layers(self):
        return [(n, list(zip(self._weight_names(n), arrs))) for n, arrs in self.get_weights_dict().items()]

    def _optimizer_state(self):
        """SGD.weights (K.optimizers.py:173-174): [iterations] + one moment per trainable weight, in parameter order"""
        ctx = self.ctx
        out = [("SGD/iterations:0", np.array(self.optimizer.iterations if self.optimizer else 0, dtype=np.int64))]
        k = 0
        for p in ctx.params:
            if not p.trainable:
                continue
            v = ctx._to_keras(p, ctx.V[p.offset:p.offset + p.numel])
            out.append(("training/SGD/Variable%s:0" % ("_%d" % k if k else ""), v))
            k += 1
        return out

    def _set_optimizer_state(self, arrays):
        """inverse of _optimizer_state: [iterations, moment...] (names are TensorFlow's auto-numbering; order decides)"""
        ctx = self.ctx
        tr = [p for p in ctx.params if p.trainable]
        if len(arrays) != len(tr) + 1:
            raise ValueError("optimizer state has %d arrays, the model needs %d" % (len(arrays), len(tr) + 1))
        its = int(np.asarray(arrays[0]).reshape(-1)[0])
        if self.optimizer is not None:
            self.optimizer.iterations = its
        else:
            self._pending_iterations = its       # compile() picks it up (Keras restores it with the optimizer, K.models.py:249-272)
        for p, a in zip(tr, arrays[1:]):
            t = torch.from_numpy(ctx._to_internal(p, a).reshape(-1)).to(ctx.dev)
            ctx.V[p.offset:p.offset + p.numel] = t

    @staticmethod
    def _is_hdf5_name(filepath):
        return str(filepath).lower().endswith((".h5", ".hdf5", ".hdf"))

    def save_weights(self, filepath, overwrite=True, nested_under=None):
        """K.engine/topology.py:2555-2588.  `.h5` / `.hdf5` names get a real Keras-layout HDF5 file (written natively by
        h5lite: the image has no h5py); other names the `.npz` container of round 1.  `nested_under='model_1'` writes
        the layout a `make_parallel` wrapper produced in the reference (every layer one level down, under the wrapped
        model's group), which is what its `by_gpu` loaders expect (topology.py:3171-3330)."""
        if not overwrite and os.path.isfile(filepath):
            raise IOError("%s exists and overwrite=False" % filepath)
        if not self._is_hdf5_name(filepath):
            return self._save_npz(filepath, with_optimizer=False)
        from . import h5lite
        root = h5lite.WGroup()
        self._fill_weights_group(root, nested_under)
        h5lite.write_file(filepath, root)

    def _fill_weights_group(self, g, nested_under=None):
        from . import h5lite
        layers = self._keras_layers()
        if nested_under:
            layers = [(nested_under, [w for _, ws in layers for w in ws])]
        h5lite.keras_weights_group(g, layers)

    def save(self, filepath, overwrite=True, include_optimizer=True):
        """K.models.py:31-170 `save_model`: model_weights group + optimizer state + training_config.  The graph topology
        is NOT serialised (model_config carries the constructor identity only): the drop-in rebuilds a model from its
        constructor, as every reference script does before load_weights."""
        if not overwrite and os.path.isfile(filepath):
            raise IOError("%s exists and overwrite=False" % filepath)
        if not self._is_hdf5_name(filepath):
            return self._save_npz(filepath, with_optimizer=include_optimizer)
        import json
        from . import h5lite
        root = h5lite.WGroup()
        root.attrs["keras_version"] = "2.0.8"
        root.attrs["backend"] = "tensorflow"
        root.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {
            "name": self.name, "constructor": {"kind": self.kind, "variant": self.variant,
                                               "input_shape": list(self.input_shape)}, "layers": []}})
        self._fill_weights_group(root.group("model_weights"))
        if include_optimizer and self.optimizer is not None:
            opt = self.optimizer
            root.attrs["training_config"] = json.dumps({
                "optimizer_config": {"class_name": "SGD", "config": {"lr": opt.lr, "momentum": opt.momentum,
                                                                     "decay": opt.decay, "nesterov": bool(opt.nesterov)}},
                "loss": list(getattr(self, "_loss_names", None) or ["weighted_crossentropy"]), "metrics": None,
                "sample_weight_mode": None, "loss_weights": None})
            og = root.group("optimizer_weights")
            st = self._optimizer_state()
            og.attrs["weight_names"] = [n for n, _ in st]
            for n, a in st:
                og.dataset(n, a)
        h5lite.write_file(filepath, root)

    def _save_npz(self, filepath, with_optimizer):
        flat = {}
        for n, arrs in self.get_weights_dict().items():
            for i, a in enumerate(arrs):
                flat["%s/%d" % (n, i)] = a
        flat["__model_name__"] = np.array(self.name or "")
        if with_optimizer and self.optimizer is not None:
            for i, (_, a) in enumerate(self._optimizer_state()):
                flat["__optimizer__/%d" % i] = a
        tmp = "%s.tmp%d" % (filepath, os.getpid())
        with open(tmp, "wb") as f:
            np.savez(f, **flat)
        os.replace(tmp, filepath)       # atomic: a reader never sees a torn file

    def load_optimizer_weights(self, filepath):
        """restore SGD iterations + momentum buffers saved by `save` (Keras does this in load_model, K.models.py:249-272)"""
        with open(filepath, "rb") as fh:
            magic = fh.read(8)
        if magic == b"\x89HDF\r\n\x1a\n":
            from . import h5lite
            f = h5lite.File(filepath)
            if "optimizer_weights" not in f:
                raise ValueError("%s holds no optimizer_weights group" % filepath)
            og = f["optimizer_weights"]
            names = [(n if isinstance(n, bytes) else bytes(n)).rstrip(b"\0").decode("utf8")
                     for n in np.asarray(og.attrs["weight_names"]).reshape(-1)]
            arrays = [np.asarray(og[n]) for n in names]
        else:
            z = np.load(filepath, allow_pickle=False)
            keys = sorted([k for k in z.files if k.startswith("__optimizer__/")], key=lambda k: int(k.split("/")[1]))
            if not keys:
                raise ValueError("%s holds no optimizer state" % filepath)
            arrays = [z[k] for k in keys]
        self._set_optimizer_state(arrays)

    def load_weights(self, filepath, by_name=False, by_gpu=False, two_model=False, by_flag=False):
        """K.engine/topology.py:2590-2640 incl. the author's loaders: `by_name` skips layers absent from the file;
        `by_name + by_gpu` reads the layers under the wrapped model's group `model_1` (:3171-3247); `+ two_model`
        under `denseu161` (by_flag) or `auto3d_residual_conv` (:3250-3330) -- the 2D DenseUNet of a multi-GPU
        pre-training run loaded into the hybrid (train_hybrid.py:146).  The default requires every layer of the model.
        A name-based load that matches NO layer raises instead of silently keeping the initial weights."""
        with open(filepath, "rb") as fh:
            magic = fh.read(8)
        lenient = bool(by_name)
        if magic == b"\x89HDF\r\n\x1a\n":
            from . import h5lite
            if by_name and by_gpu:
                grp = ("denseu161" if by_flag else "auto3d_residual_conv") if two_model else "model_1"
                d = h5lite.read_nested_model_weights(filepath, grp, swap="len2or4" if two_model else "always")
            else:
                d = OrderedDict((n, arrs) for n, arrs in h5lite.read_keras_weights(filepath).items() if arrs)
            extra = [n for n in d if n not in self.ctx.by_layer]
            if extra and not lenient:
                raise ValueError("weights file has layers the model lacks: %s... (use by_name=True)" % extra[:5])
        else:
            z = np.load(filepath, allow_pickle=False)
            groups = OrderedDict()
            for k in z.files:
                if k.startswith("__"):
                    continue
                n, i = k.rsplit("/", 1)
                groups.setdefault(n, {})[int(i)] = z[k]
            d = OrderedDict((n, [g[i] for i in sorted(g)]) for n, g in groups.items())
        if not lenient:
            missing = [n for n in self.ctx.by_layer if n not in d]
            if missing:
                raise ValueError("weights file lacks layers: %s..." % missing[:5])
        elif not any(n in self.ctx.by_layer for n in d):
            raise ValueError("%s: none of its %d layers (%s...) matches a layer of model %s -- nothing would be loaded; for "
                             "a checkpoint written from a make_parallel wrapper pass by_name=True, by_gpu=True"
                             % (filepath, len(d), list(d)[:3], self.name))
        self.set_weights_dict(d, strict=not lenient)

    def count_params(self):
        return int(sum(int(np.prod(p.keras_shape)) for p in self.ctx.params))

    def summary(self):
        print("Model %s (%s, %s)" % (self.name, self.kind, "bf16" if self.dtype == HDU_BF16 else "f32"))
        for n, ps in self.ctx.by_layer.items():
            print("  %-28s %s" % (n, [p.keras_shape for p in ps]))
        print("Total params: %d" % self.count_params())


class ModelCheckpoint:
    """K.callbacks.py:335-432 subset: save every `period` epochs to filepath.format(epoch=, **logs)."""

    def __init__(self, filepath, monitor="loss", verbose=0, save_best_only=False, save_weights_only=False, mode="min",
                 period=1):
        self.filepath, self.monitor, self.verbose, self.save_best_only, self.period = filepath, monitor, verbose, save_best_only, period
        self.save_weights_only = save_weights_only
        self.best = np.inf
        self.model = None
        self.epochs_since = 0

    def set_model(self, model):
        self.model = model

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.epochs_since += 1
        if self.epochs_since < self.period:
            return
        self.epochs_since = 0
        cur = logs.get(self.monitor, np.inf)
        if self.save_best_only and not cur < self.best:
            return
        self.best = min(self.best, cur)
        path = self.filepath.format(epoch=epoch, **logs)      # Keras 2.0.8 formats the 0-based epoch (K.callbacks.py:404)
        if not _is_io_rank():
            return            # every rank holds the same weights after the all-reduced step; rank 0 writes the file
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if self.save_weights_only:
            self.model.save_weights(path)      # K.callbacks.py:425-428
        else:
            self.model.save(path)


def make_parallel(model, gpu_count, mini_batch=None):
    """K.utils2/multi_gpu.py:7-69 builds in-graph towers in ONE process.  The MI355X design is one process per GPU
    (launched by torch.distributed.run): inside such a process this binds the model to the RCCL gradient all-reduce
    and returns it; with gpu_count <= 1 it is the identity (the reference's gpu_count=0 case, train_2ddense.py:180)."""
    from . import parallel
    if gpu_count is None or gpu_count <= 1:
        return model
    parallel.attach_data_parallel(model, expected_world=gpu_count)
    return model
