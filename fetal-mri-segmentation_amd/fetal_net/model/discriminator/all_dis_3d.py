"""`discriminator_image_3d` with the reference signature and topology (reference fetal_net/model/discriminator/all_dis_3d.py:11-72),
executed by the layer-graph engine (fmri_hip.graph_engine) with a Dense(1, 'sigmoid') head and the binary cross-entropy loss.

conv_block(level): [Conv3D(3x3x3, 'same') -> InstanceNormalization(axis=1) -> LeakyReLU] -> SpatialDropout3D -> [same trio] ->
AveragePooling3D(); the first block's first conv has strides (2, 2, 1) (the patches are thin along Z), the others stride 1; the filter
count doubles per level up to 128.  The stack stops early once the second-to-last axis is shorter than the kernel, and every level it
did not build becomes a Dense(128, LeakyReLU) after the GlobalAveragePooling3D; Dense(1, 'sigmoid') is the output.
Compiled with Adam(lr, beta_1 = 0.5), loss = mean binary cross-entropy over the flattened batch, metric 'mae'.
"""
from ...engine_model import Adam
from ..graph import Graph


def _mini_conv_block(g, x, n_filters, kernel, strides=1):
    h = g.conv(x, n_filters, kernel, strides=strides, padding='same')
    h = g.instance_norm(h, axis=1)
    return g.leaky_relu(h)


def _conv_block(g, x, level, n_base_filters, kernel, strides, dropout_rate):
    n_filters = min(128, (2 ** level) * n_base_filters)
    nd = len(kernel)
    h = _mini_conv_block(g, x, n_filters, kernel, strides)
    h = g.spatial_dropout(h, dropout_rate, nd)
    h = _mini_conv_block(g, h, n_filters, kernel)
    return g.avg_pool(h, (2,) * nd)


def d_loss(y_true, y_pred):
    """binary_crossentropy(K.batch_flatten(y_true), K.batch_flatten(y_pred)) (reference all_dis_3d.py:47-50); host evaluation"""
    import numpy as np
    from ...metrics import binary_crossentropy
    yt, yp = np.asarray(y_true, np.float64), np.asarray(y_pred, np.float64)
    return binary_crossentropy(yt.reshape(yt.shape[0], -1), yp.reshape(yp.shape[0], -1))


def discriminator_image_3d(input_shape=(None, 2, 64, 128, 128), n_base_filters=16, optimizer=Adam, initial_learning_rate=5e-4, depth=5,
                           dropout_rate=0.3, **kargs):
    from ...adversarial import DiscriminatorModel
    input_shape = tuple(input_shape)
    if len(input_shape) != 4 or any(v is None for v in input_shape):
        # the reference's default (None, 2, 64, 128, 128) is not a valid Input shape for a Conv3D stack either: callers pass
        # [channels + n_labels, X, Y, Z] (reference train_adv.py:145-149)
        raise ValueError("Input 0 is incompatible with layer conv3d_1: expected ndim=5, found ndim=%d" % (len(input_shape) + 1))
    input_shape = tuple(int(v) for v in input_shape)
    kernel, scale_only_xy = (3, 3, 3), 1
    g = Graph()
    cur = g.input(input_shape)
    fc_layers = 0
    for level in range(scale_only_xy):
        cur = _conv_block(g, cur, level, n_base_filters, kernel, (2, 2, 1), dropout_rate)
    for level in range(scale_only_xy, depth):
        cur = _conv_block(g, cur, level, n_base_filters, kernel, 1, dropout_rate)
        if cur.shape[-2] < kernel[0]:
            fc_layers = depth - level - 1
            break
    cur = g.global_avg_pool(cur)
    for _ in range(fc_layers):
        cur = g.dense(cur, 128, activation="leaky_relu")
    g.dense(cur, 1, activation="sigmoid")
    for l in g.layers:
        if l.class_name == "SpatialDropout3D":
            l.config["data_format"] = "channels_first"
    builder_kwargs = dict(input_shape=input_shape, n_base_filters=n_base_filters, initial_learning_rate=initial_learning_rate, depth=depth,
                          dropout_rate=dropout_rate)
    if "compute_dtype" in kargs:
        builder_kwargs["compute_dtype"] = kargs["compute_dtype"]
    model = DiscriminatorModel(g.layers, None, "discriminator_image_3d", builder_kwargs, "channels_first_3d", name="Discriminator")
    model.compile(optimizer=optimizer(lr=initial_learning_rate, beta_1=0.5), loss=d_loss, metrics=['mae'])
    return model
