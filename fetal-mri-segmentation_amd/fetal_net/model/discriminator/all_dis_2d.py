"""`discriminator_image_2d` (reference fetal_net/model/discriminator/all_dis_2d.py:11-48).

The reference builder hands Conv2D the 3-tuple strides (2, 2, 1) of its 3-D twin (all_dis_2d.py:31-32), which Keras rejects while the
first layer is being constructed: the function cannot return a model there.  The mirror keeps the signature and fails the same way,
with Keras' message, instead of inventing a topology the reference never ran.
"""
from ...engine_model import Adam
from ..graph import Graph


def discriminator_image_2d(input_shape=(None, 2, 64, 128, 128), n_base_filters=16, optimizer=Adam, initial_learning_rate=5e-4, depth=5,
                           dropout_rate=0.3, **kargs):
    g = Graph()
    x = g.input(tuple(input_shape))
    g.conv(x, min(128, n_base_filters), (3, 3), strides=(2, 2, 1), padding='same')       # raises: strides must be a tuple of 2 integers
    raise AssertionError("unreachable")
