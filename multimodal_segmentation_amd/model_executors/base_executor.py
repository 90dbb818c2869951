"""Base class of the executors (reference model_executors/base_executor.py): batch alignment, the background
("residual") mask channel, and the augmenting data generators (keras ImageDataGenerator with +-20 degree rotations,
base_executor.py:37-78,103-110 -> utils/augment.RotationFlow: shuffle, gather and rotate on the device)."""
import logging

import numpy as np

log = logging.getLogger('executor')


class Executor(object):
    def __init__(self, conf, model):
        self.conf = conf
        self.model = model
        self.loader = model.loader
        self.epoch = 0
        self.batch = 0
        self.batches = 0

    def add_residual(self, data):
        """reference base_executor.py:83-87"""
        residual = np.ones(data.shape[:-1] + (1,))
        for i in range(data.shape[-1]):
            residual[data[..., i:i + 1] == 1] = 0
        return np.concatenate([data, residual], axis=-1)

    def align_batches(self, array_list):
        """reference base_executor.py:112-119"""
        mn = np.min([x.shape[0] for x in array_list])
        return [x[0:mn] + 0. for x in array_list]

    def get_datagen_params(self):
        """augmentation switches of base_executor.py:103-110: only rotation_range is non-trivial"""
        return dict(horizontal_flip=False, vertical_flip=False, rotation_range=20., width_shift_range=0, height_shift_range=0,
                    zoom_range=0)

    def get_data_generator(self, train_images=None, train_labels=None):
        """Iterator over (images..., labels...) batches, all shuffled and rotated identically (base_executor.py:37-78: one
        keras flow per array with a shared seed, zipped).  A single array yields bare batches instead of 1-tuples."""
        from ..utils.augment import RotationFlow
        arrays = []
        for group in (train_images, train_labels):
            if group is not None:
                arrays += list(group) if isinstance(group, (list, tuple)) else [group]
        if not arrays:
            raise Exception('No data to iterate.')
        p = self.get_datagen_params()
        unsupported = [k for k in ('horizontal_flip', 'vertical_flip', 'width_shift_range', 'height_shift_range', 'zoom_range')
                       if p.get(k)]
        if unsupported:
            raise NotImplementedError('augmentations not on the device path: %s' % unsupported)
        # data parallel: rank r reseeds the global numpy stream with conf.seed + 7919 r (+ batch index), so shuffles, rotation
        # angles and the z / pool-index draws that follow a batch differ between replicas; rank 0 follows the reference stream
        from ..parallel import dp
        return RotationFlow(arrays, self.conf.batch_size, self.conf.seed + 7919 * dp.rank(), self.device,
                            rotation_range=p['rotation_range'],
                            order=self.conf.get('augment_interpolation_order', 1))

    def validate(self, epoch_loss):
        pass

    def stop_criterion(self, es, logs):
        es.on_epoch_end(self.epoch, logs)
        return es.stopped_epoch > 0


class EarlyStopping(object):
    """keras.callbacks.EarlyStopping(monitor, min_delta, patience), mode 'min' (dafnet_executor.py:222)"""

    def __init__(self, monitor, min_delta=0., patience=0):
        self.monitor, self.min_delta, self.patience = monitor, min_delta, patience
        self.best = np.inf
        self.wait = 0
        self.stopped_epoch = 0

    def on_epoch_end(self, epoch, logs):
        cur = logs.get(self.monitor)
        if cur is None:
            return
        if cur < self.best - self.min_delta:
            self.best, self.wait = cur, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
