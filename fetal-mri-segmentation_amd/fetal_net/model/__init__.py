"""Builders looked up by name: `getattr(fetal_net.model, config['model_name'])` (reference fetal/train_fetal.py:32,
fetal_net/model/__init__.py:3-18).  Only the hot-path builders are provided."""
from .unet3d.unet import unet_model_3d
from .unet3d.isensee2017 import isensee2017_model_3d
from .unet.unet import unet_model_2d


def _not_yet(name, anchor):
    def f(*a, **k):
        raise NotImplementedError("%s is a later row of the hot-path scope table (SURVEY.md §8a, reference %s)" % (name, anchor))
    f.__name__ = name
    return f


isensee2017_model = _not_yet("isensee2017_model", "fetal_net/model/unet/isensee.py")
