"""Modality (intensity) encoder: (anatomy, image) -> (z, KL) (reference model_components/modality_encoder.py:13-52).

4 x [conv3x3 stride 2 valid + LeakyReLU(0.3)] -> Flatten -> Dense 32 + LeakyReLU -> z_mean, z_log_var (Dense num_z)
-> z = mean + exp(0.5 log_var) * eps (utils/sdnet_utils.py:9-21), KL (costs.py:186-189).  The Concatenate of the
9 input channels is folded into the first convolution's gather.  `mu_only=True` is the reference's Enc_Modality_mu
sub-model cut at layer `z_mean` (models/dafnet.py:126).
"""
import logging

import numpy as np

from .. import graphs, nn, ops
from ..utils.rng import global_rng

log = logging.getLogger('modality_encoder')


class ModalityEncoder(nn.Model):
    def __init__(self, conf, rng):
        super(ModalityEncoder, self).__init__('Enc_Modality')
        self.conf = conf
        H, W, sc = conf.anatomy_encoder.output_shape
        c = sc + conf.input_shape[-1]
        for i, f in enumerate((16, 32, 64, 128)):
            nn.conv_params(self, 'c%d' % i, 3, c, f, 'he_normal')
            c = f
            H, W = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        nn.dense_params(self, 'd0', H * W * c, 32, 'he_normal')
        nn.dense_params(self, 'z_mean', 32, conf.num_z)
        nn.dense_params(self, 'z_log_var', 32, conf.num_z)
        self.finalize(rng)
        self.output_shape = [(None, conf.num_z), (None, 1)]
        self._eps_seed = int(conf.get('seed', 10)) + 104729
        self._eps_rng = None

    def _mean_logvar(self, s, x, want_logvar=True):
        l = nn.conv(self, 'c0', s, stride=2, padding='valid', act='leaky', alpha=0.3, x2=x)
        for i in (1, 2, 3):
            l = nn.conv(self, 'c%d' % i, l, stride=2, padding='valid', act='leaky', alpha=0.3)
        l = nn.dense(self, 'd0', l.reshape(l.shape[0], -1), act='leaky', alpha=0.3)
        # Enc_Modality_mu is the same network cut at z_mean (dafnet.py:126): the log-variance head is not evaluated there
        if not want_logvar:
            return nn.dense(self, 'z_mean', l), None
        l = ops.Shared(l, 2)           # both heads read it: their gradients are added by one launch in the backward pass
        return nn.dense(self, 'z_mean', l.use()), nn.dense(self, 'z_log_var', l.use())

    def forward(self, s, x, training=False, eps=None, mu_only=False):
        z_mean, z_log_var = self._mean_logvar(s, x, want_logvar=not mu_only)
        if mu_only:
            return z_mean
        if eps is None:
            # K.random_normal inside the graph (sdnet_utils.py:20) is TensorFlow's generator, independent of numpy's global
            # stream (from which the executors draw z samples and pool indices right after a batch): a private generator,
            # created at the first draw and offset by the data-parallel rank so that replicas sample different noise
            if self._eps_rng is None:
                from ..parallel import dp
                self._eps_rng = np.random.RandomState((self._eps_seed + 7919 * dp.rank()) % (2 ** 32))
            shape = tuple(z_mean.shape)
            eps = graphs.host_draw(lambda: self._eps_rng.normal(0., 1., size=shape).astype(np.float32), z_mean.device)
        eps = nn.to_device(eps, z_mean.device)
        z, kl = ops.sampling_kl(z_mean, z_log_var, eps)
        return [z, kl]


def build(conf, rng=None):
    """Build an encoder to extract intensity information from the image."""
    model = ModalityEncoder(conf, rng or global_rng())
    log.info('Enc_Modality')
    model.summary(print_fn=log.debug)
    return model
