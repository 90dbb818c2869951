"""Stochastic Weight Averaging (reference callbacks/swa.py:14-47): up to `swa_epoch` the SWA weights track the live
weights; afterwards they are the running average  swa <- (swa*(e - s) + w) / ((e - s) + 1)  over all weights returned by
get_weights() (BatchNorm moving statistics included, as in the reference).  The averages live on the device and are
updated with the axpby kernel; `get_clone_model()` builds (once) a second model through the same build function and loads
the averaged weights into it -- validation, testing and checkpoints use these clones (dafnet_executor.py:286-335)."""
import logging


from .. import ops

log = logging.getLogger('swa')


class SWA(object):
    def __init__(self, swa_epoch, model_build_fnc, build_params):
        self.swa_epoch = swa_epoch
        self.model_build_fnc = model_build_fnc
        self.build_params = build_params
        self.clone = None
        self.model = None
        self.swa_weights = None
        self.params = {'epochs': None}

    def on_train_begin(self, logs=None):
        self.nb_epoch = self.params.get('epochs')

    def on_epoch_end(self, epoch, logs=None):
        live = [p.data for p in self.model.all_params()]
        if epoch <= self.swa_epoch or self.swa_weights is None:
            if self.swa_weights is None:
                self.swa_weights = [w.clone() for w in live]
            else:
                for s, w in zip(self.swa_weights, live):
                    s.copy_(w)
        else:
            k = float(epoch - self.swa_epoch)
            for s, w in zip(self.swa_weights, live):
                ops.axpby(s, w, k / (k + 1.0), 1.0 / (k + 1.0), out=s)

    def on_train_end(self, logs=None):
        for p, s in zip(self.model.all_params(), self.swa_weights):
            p.data.copy_(s)
        ops.bump_weight_version()
        log.debug('Final model parameters set to stochastic weight average.')

    def get_clone_model(self):
        if self.clone is None:
            if self.build_params is not None:
                self.clone = self.model_build_fnc(self.build_params)
            else:
                self.clone = self.model_build_fnc()
        src = self.swa_weights if self.swa_weights is not None else [p.data for p in self.model.all_params()]
        dst = self.clone.all_params()
        assert len(src) == len(dst), 'clone of %s has %d weights, live model %d' % (self.model.name, len(dst), len(src))
        for p, s in zip(dst, src):
            assert tuple(p.shape) == tuple(s.shape), (p.name, p.shape, tuple(s.shape))
            p.data.copy_(s)
        ops.bump_weight_version()
        return self.clone
