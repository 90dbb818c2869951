"""A compiled Keras "trainer" model: a forward graph over shared component models + a per-output loss table + one
Adam state.  `fit(inputs, targets)` performs exactly what `keras.Model.fit(..., epochs=1)` does for one batch
(batch_size 32 >= the batch, so one optimiser step): forward in training mode (BatchNorm batch statistics, moving
averages updated), weighted sum of the per-output losses (+ the layers' regulariser losses), gradients of the
trainable weights collected at compile time, one Adam update, History with '<output_name>_loss' entries.

Loss kinds (reference costs.py / keras): 'dice_bce' (make_combined_dice_bce), 'dice' (make_dice_loss_fnc), 'mse',
'mae', 'ypred' (mean of the output).
"""
import torch

from .. import graphs, nn, ops
from ..parallel import dp


class OutputSpec(object):
    __slots__ = ('name', 'kind', 'weight')

    def __init__(self, name, kind, weight):
        self.name, self.kind, self.weight = name, kind, float(weight)


class Trainer(object):
    def __init__(self, name, graph_fn, output_specs, train_models, optimizer, num_masks=None, regularised=(),
                 all_models=()):
        """graph_fn(inputs: list of device tensors) -> list of output tensors (len == len(output_specs)).
        train_models: nn.Models whose arenas this trainer updates (the weights that were trainable at compile time).
        regularised: DiscriminatorModels whose Spectral penalties are part of this trainer's total loss."""
        self.name = name
        self.graph_fn = graph_fn
        self.specs = list(output_specs)
        self.train_models = list(train_models)
        self.optimizer = optimizer
        self.num_masks = num_masks
        self.regularised = list(regularised)
        self.all_models = list(all_models) or self.train_models
        self.device = self.train_models[0].device
        self.last_outputs = None
        # conf.hip_graphs: record the step into a hipGraph after two eager warm-up steps and replay it from then on (graphs.py)
        self.use_graph = False
        self._graphs = {}
        # static loss scale (fp16 compute mode): every seed gradient is multiplied by it, the gradient arenas are divided by it
        # before the optimiser step; 1.0 = off
        self.loss_scale = 1.0
        # (compute_dtype, 16-bit activation storage) of the model wrapper this trainer belongs to: every fit / predict runs inside
        # ops.precision_scope(self.precision); None = whatever the library is set to
        self.precision = None

    # keras API used by the executors ---------------------------------------------------------------------------
    @property
    def output_names(self):
        return [s.name for s in self.specs]

    def summary(self, print_fn=print):
        print_fn('Trainer %s: %d outputs, trainable models: %s' % (self.name, len(self.specs),
                                                                   ', '.join(m.name for m in self.train_models)))

    def _loss_and_grad(self, spec, pred, target):
        gs = spec.weight * self.loss_scale            # scale of the returned gradient only; the loss value is unscaled
        if spec.kind == 'dice_bce':
            return ops.seg_loss(pred, target, self.num_masks, 0.01, gs, class_sum_hook=dp.class_sum_hook(),
                                n_pix_global=pred.numel() // pred.shape[-1] * dp.world_size())
        if spec.kind == 'dice':
            return ops.seg_loss(pred, target, self.num_masks, 0.0, gs)
        if spec.kind == 'mse':
            return ops.diff_loss(pred, target, 'mse', gs)
        if spec.kind == 'mae':
            return ops.diff_loss(pred, target, 'mae', gs)
        if spec.kind == 'ypred':
            return ops.diff_loss(pred, 0.0, 'mean', gs)
        raise ValueError(spec.kind)

    def _prep_target(self, t, pred):
        if isinstance(t, (int, float)):
            return float(t)
        t = nn.to_device(t, self.device)
        if t.numel() != pred.numel():
            # keras broadcasts e.g. zeros(batch) targets against [batch, 1] outputs
            t = t.reshape(-1, *([1] * (pred.dim() - 1))).expand_as(pred).contiguous()
        return t

    def fit(self, inputs, targets, epochs=1, verbose=0, **graph_kw):
        assert epochs == 1
        inputs = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        targets = list(targets) if isinstance(targets, (list, tuple)) else [targets]
        graph_kw = {k: v for k, v in graph_kw.items() if v is not None}      # (eps=None etc.: the graph functions' defaults)
        with ops.precision_scope(self.precision):
            if self.use_graph and not graph_kw and not dp.enabled() and self.device.type == 'cuda':
                key = graphs.signature(inputs, targets)
                st = self._graphs.get(key)
                if st is None:
                    st = self._graphs[key] = graphs.FitGraph(self)
                return st.run(inputs, targets)
            return self._fit_eager(inputs, targets, graph_kw)

    @staticmethod
    def _total(terms):
        return nn_total(terms)

    def _fit_eager(self, inputs, targets, graph_kw, lr_dev=None):
        ins = [nn.to_device(x, self.device) for x in inputs]
        assert len(targets) == len(self.specs), '%s: %d targets for %d outputs' % (self.name, len(targets), len(self.specs))
        for m in self.train_models:
            m.zero_grad_own()
        # data parallel: arenas are all-reduced as soon as their last gradient kernel is queued (overlap with backward);
        # models that also receive regulariser gradients after the backward pass are reduced at the end
        tracker = dp.begin(self.train_models, defer=[d for d in self.regularised if d in self.train_models])
        with torch.enable_grad():
            outs = self.graph_fn(ins, **graph_kw)
            assert len(outs) == len(self.specs)
            hist = nn.History()
            grads, terms = [], []
            for spec, pred, tgt in zip(self.specs, outs, targets):
                loss, dpred = self._loss_and_grad(spec, pred, self._prep_target(tgt, pred))
                grads.append(dpred)
                terms.append((spec.weight, loss))
                hist.record(spec.name + '_loss', loss)
            torch.autograd.backward(outs, grads)
        for d in self.regularised:
            for loss in d.regulariser_losses(accumulate_grad=d in self.train_models, grad_scale=self.loss_scale):
                terms.append((1.0, loss))
        dp.finish(tracker)
        if self.loss_scale != 1.0:
            for m in self.train_models:
                ops.axpby(m.grad_arena, m.grad_arena, 1.0 / self.loss_scale, 0.0, out=m.grad_arena)
        self.optimizer.step(self.train_models, lr_dev=lr_dev)
        hist.record('loss', nn_total(terms))
        self.last_outputs = [o.detach() for o in outs]
        return hist

    def predict(self, inputs, **graph_kw):
        ins = [nn.to_device(x, self.device) for x in (inputs if isinstance(inputs, (list, tuple)) else [inputs])]
        with torch.no_grad(), ops.precision_scope(self.precision):
            outs = self.graph_fn(ins, training=False, **graph_kw)
        return [nn.to_numpy(o) for o in outs]


class nn_total(object):
    """Lazy weighted sum of device scalars (read with .item(), like a 0-d tensor)."""

    def __init__(self, terms):
        self.terms = terms

    def item(self):
        return float(sum(w * float(t.item()) for w, t in self.terms))
