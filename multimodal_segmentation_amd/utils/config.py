"""EasyDict stand-in (the reference wraps its config dicts in easydict.EasyDict, experiment.py:41)."""


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super(EasyDict, self).__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super(EasyDict, self).__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v
