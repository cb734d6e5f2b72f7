"""Tiny symbolic layer recorder: gives the builders Keras-style auto names (`conv3d_3`, `batch_normalization_3`, ...)
with per-class counters in creation order, output shapes and parameter counts — what reference callers inspect
(`model.layers[i].name`, `model.output_shape`, `model.summary()`; reference test/test_model.py:11-15)."""
from ..engine_model import Layer


class Sym(object):
    def __init__(self, name, shape):
        self.name, self.shape = name, tuple(shape)
        self._keras_shape = self.shape


class Graph(object):
    def __init__(self):
        self.layers, self._count = [], {}

    def _name(self, base):
        self._count[base] = self._count.get(base, 0) + 1
        return "%s_%d" % (base, self._count[base])

    def _add(self, base, cls, shape, inbound, **config):
        name = self._name(base)
        self.layers.append(Layer(name, cls, shape, config, [s.name for s in inbound]))
        return Sym(name, shape)

    def input(self, shape):
        return self._add("input", "InputLayer", (None,) + tuple(shape), [])

    def conv(self, x, filters, kernel, strides=None, padding="valid"):
        nd = len(kernel)
        if isinstance(strides, int):
            strides = (strides,) * nd
        strides = tuple(strides or (1,) * nd)
        if len(strides) != nd:          # keras.utils.conv_utils.normalize_tuple
            raise ValueError('The `strides` argument must be a tuple of %d integers. Received: %s' % (nd, str(strides)))
        sp = [(-(-d // s) if padding == "same" else (d - k) // s + 1) for d, s, k in zip(x.shape[2:], strides, kernel)]
        params = int(filters * x.shape[1] * _prod(kernel) + filters)
        return self._add("conv%dd" % nd, "Conv%dD" % nd, (None, filters) + tuple(sp), [x], filters=filters, kernel_size=tuple(kernel),
                         strides=strides, padding=padding, params=params)

    def deconv(self, x, filters, kernel, strides):
        nd = len(kernel)
        sp = [d * s for d, s in zip(x.shape[2:], strides)]
        params = int(filters * x.shape[1] * _prod(kernel) + filters)
        return self._add("conv%dd_transpose" % nd, "Conv%dDTranspose" % nd, (None, filters) + tuple(sp), [x], filters=filters,
                         kernel_size=tuple(kernel), strides=tuple(strides), params=params)

    def batch_norm(self, x, axis=1):
        return self._add("batch_normalization", "BatchNormalization", x.shape, [x], axis=axis, params=4 * x.shape[axis])

    def instance_norm(self, x, axis=1):
        return self._add("instance_normalization", "InstanceNormalization", x.shape, [x], axis=axis, params=2 * x.shape[axis])

    def activation(self, x, name):
        return self._add("activation", "Activation", x.shape, [x], activation=name)

    def leaky_relu(self, x, alpha=0.3):
        return self._add("leaky_re_lu", "LeakyReLU", x.shape, [x], alpha=alpha)

    def max_pool(self, x, size):
        nd = len(size)
        return self._add("max_pooling%dd" % nd, "MaxPooling%dD" % nd, x.shape[:2] + tuple(d // p for d, p in zip(x.shape[2:], size)), [x],
                         pool_size=tuple(size))

    def avg_pool(self, x, size):
        nd = len(size)
        return self._add("average_pooling%dd" % nd, "AveragePooling%dD" % nd, x.shape[:2] + tuple(d // p for d, p in zip(x.shape[2:], size)), [x],
                         pool_size=tuple(size))

    def global_avg_pool(self, x):
        nd = len(x.shape) - 2
        return self._add("global_average_pooling%dd" % nd, "GlobalAveragePooling%dD" % nd, (None, x.shape[1]), [x])

    def dense(self, x, units, activation=None):
        """activation: None | 'sigmoid' | 'leaky_relu' (a keras.layers.LeakyReLU() instance handed to Dense: the instance takes a
        leaky_re_lu_<n> name from the counter although it never becomes a node of the graph)"""
        alpha = None
        if activation == "leaky_relu":
            self._name("leaky_re_lu")
            alpha = 0.3
        return self._add("dense", "Dense", (None, units), [x], units=units, activation=activation, alpha=alpha, params=int(x.shape[1] * units + units))

    def up_sample(self, x, size):
        nd = len(size)
        return self._add("up_sampling%dd" % nd, "UpSampling%dD" % nd, x.shape[:2] + tuple(d * p for d, p in zip(x.shape[2:], size)), [x],
                         size=tuple(size))

    def concat(self, xs, axis=1):
        shape = list(xs[0].shape)
        shape[axis] = sum(s.shape[axis] for s in xs)
        return self._add("concatenate", "Concatenate", shape, xs, axis=axis)

    def add(self, xs):
        return self._add("add", "Add", xs[0].shape, xs)

    def permute(self, x, dims):
        return self._add("permute", "Permute", (None,) + tuple(x.shape[d] for d in dims), [x], dims=tuple(dims))

    def spatial_dropout(self, x, rate, nd):
        return self._add("spatial_dropout%dd" % nd, "SpatialDropout%dD" % nd, x.shape, [x], rate=rate)


def _prod(t):
    p = 1
    for v in t:
        p *= int(v)
    return p
