"""Base class of the model wrappers (reference models/basenet.py:17-38)."""
from ..loaders import loader_factory


class BaseNet(object):
    def __init__(self, conf):
        self.model = None
        self.conf = conf
        self.loader = None
        if hasattr(self.conf, 'dataset_name') and len(self.conf.dataset_name) > 0:
            self.loader = loader_factory.init_loader(self.conf.dataset_name)

    def build(self):
        pass

    def load_models(self):
        pass
