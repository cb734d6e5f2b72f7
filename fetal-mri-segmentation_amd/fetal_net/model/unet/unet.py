"""`unet_model_2d` with the reference signature and topology (reference fetal_net/model/unet/unet.py:22-141): the 2-D twin of
unet_model_3d with channels-LAST input (X, Y, C) wrapped in Permute layers, optional SpatialDropout2D."""
from ...engine_model import Adam, Model
from ...metrics import dice_coefficient, dice_coefficient_loss, vod_coefficient
from ..graph import Graph


def _block(g, x, n_filters, batch_normalization):
    h = g.conv(x, n_filters, (3, 3), strides=(1, 1), padding='same')
    if batch_normalization:
        h = g.batch_norm(h, axis=1)
    return g.activation(h, 'relu')


def unet_model_2d(input_shape, pool_size=(2, 2), n_labels=1, initial_learning_rate=0.00001, deconvolution=False, depth=4,
                  n_base_filters=32, include_label_wise_dice_coefficients=False, batch_normalization=False,
                  activation_name="sigmoid", loss_function=dice_coefficient_loss, dropout_rate=0, **kargs):
    input_shape = tuple(int(v) for v in input_shape)
    pool_size = tuple(pool_size)
    g = Graph()
    x = g.input(input_shape)
    h = g.permute(x, (3, 1, 2))
    skips = []
    for level in range(depth):
        h = _block(g, h, n_base_filters * (2 ** level), batch_normalization)
        if dropout_rate > 0:
            h = g.spatial_dropout(h, dropout_rate, 2)
        h = _block(g, h, n_base_filters * (2 ** level) * 2, batch_normalization)
        skips.append(h)
        if level < depth - 1:
            h = g.max_pool(h, pool_size)
    for level in range(depth - 2, -1, -1):
        up = g.deconv(h, h.shape[1], (2, 2), (2, 2)) if deconvolution else g.up_sample(h, pool_size)
        cat = g.concat([up, skips[level]], axis=1)
        h = _block(g, cat, skips[level].shape[1], batch_normalization)
        if dropout_rate > 0:
            h = g.spatial_dropout(h, dropout_rate, 2)
        h = _block(g, h, skips[level].shape[1], batch_normalization)
    h = g.conv(h, n_labels, (1, 1))
    h = g.activation(h, activation_name)
    g.permute(h, (2, 3, 1))
    builder_kwargs = dict(input_shape=input_shape, pool_size=pool_size, n_labels=n_labels, initial_learning_rate=initial_learning_rate,
                          deconvolution=deconvolution, depth=depth, n_base_filters=n_base_filters,
                          batch_normalization=batch_normalization, activation_name=activation_name, loss_function=loss_function,
                          dropout_rate=dropout_rate)
    if "compute_dtype" in kargs:
        builder_kwargs["compute_dtype"] = kargs["compute_dtype"]
    plan_args = dict(in_channels=input_shape[-1], spatial=input_shape[:2], depth=depth, n_base_filters=n_base_filters,
                     n_labels=n_labels, ndim=2, norm="batch" if batch_normalization else None, deconvolution=bool(deconvolution))
    model = Model(g.layers, plan_args, "unet_model_2d", builder_kwargs, "channels_last_2d")
    unsupported = []
    if pool_size != (2, 2):
        unsupported.append("pool_size != (2,2)")
    if activation_name != "sigmoid":
        unsupported.append("activation_name != 'sigmoid'")
    if dropout_rate > 0:
        unsupported.append("dropout_rate > 0 (SpatialDropout2D)")
    if unsupported:
        model._unsupported = ", ".join(unsupported)
    metrics = ['binary_accuracy', vod_coefficient]
    if loss_function != dice_coefficient_loss:
        metrics += [dice_coefficient]
    model.compile(optimizer=Adam(lr=initial_learning_rate), loss=loss_function, metrics=metrics)
    return model
