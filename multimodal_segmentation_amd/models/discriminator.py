"""DCGAN discriminator with the Spectral regulariser and LS-GAN loss (reference models/discriminator.py:16-41,
layers/spectralnorm.py:199-239)."""
from .. import nn, ops
from ..utils.rng import global_rng


class DiscriminatorModel(nn.Model):
    def __init__(self, conf, rng):
        super(DiscriminatorModel, self).__init__(conf.name)
        inp_shape = tuple(conf.input_shape)
        f = conf.filters
        self.downsample_blocks = 3 if not hasattr(conf, 'downsample_blocks') else conf.downsample_blocks
        assert self.downsample_blocks > 1, self.downsample_blocks
        H, W, c = inp_shape
        nn.conv_params(self, 'c0', 4, c, f, 'he_normal')
        H, W = (H - 4) // 2 + 1, (W - 4) // 2 + 1
        c = f
        self.strides = []
        for i in range(self.downsample_blocks):
            s = 1 if i == self.downsample_blocks - 1 else 2
            cout = f * 2 * (2 ** i)
            nn.conv_params(self, 'c%d' % (i + 1), 4, c, cout, 'he_normal')
            # Spectral(dim = Cin*4*4, alpha = 10): u ~ U(-1, 1) drawn ONCE at construction and never updated
            # (spectralnorm.py:213,228-234)
            self.add_param('c%d/u0' % (i + 1), (c * 16, 1), 'uniform_pm1', trainable=False)
            self.strides.append(s)
            c = cout
            H, W = (H - 4) // s + 1, (W - 4) // s + 1
        if H < 1 or W < 1:
            raise ValueError('input %s too small for the discriminator' % (inp_shape,))
        nn.dense_params(self, 'out', H * W * c, 1)
        self.finalize(rng)
        self.input_shape = (None,) + inp_shape
        self.output_shape = (None, 1)

    def forward(self, x, training=False):
        l = nn.conv(self, 'c0', x, stride=2, padding='valid', act='leaky', alpha=0.2)
        for i, s in enumerate(self.strides):
            l = nn.conv(self, 'c%d' % (i + 1), l, stride=s, padding='valid', act='leaky', alpha=0.2)
        return nn.dense(self, 'out', l.reshape(l.shape[0], -1))

    def regulariser_losses(self, accumulate_grad=True, grad_scale=1.0):
        """Sum of the Spectral penalties of the down-sample blocks; their gradient is accumulated into the gradient
        arena when the model is trainable.  -> list of device scalars."""
        names = ['c%d' % (i + 1) for i in range(len(self.strides))]
        out = []
        for g0 in range(0, len(names), 4):            # up to 4 kernels per batch of launches
            grp = names[g0:g0 + 4]
            ws_ = [self.params[n + '/kernel'] for n in grp]
            loss, sgn = ops.spectral_reg_multi([w.data for w in ws_], [self.params[n + '/u0'].data for n in grp], 10.0)
            if accumulate_grad and self.trainable:
                ops.spectral_reg_grad_accumulate([w.data for w in ws_], sgn, [w.grad for w in ws_], scale=grad_scale)
            out += [loss[i:i + 1] for i in range(len(grp))]
        return out


class Discriminator(object):
    """reference API: D = Discriminator(conf); D.build(); D.model"""

    def __init__(self, conf):
        self.conf = conf
        self.model = None

    def build(self, rng=None):
        self.model = DiscriminatorModel(self.conf, rng or global_rng())
        return self.model
