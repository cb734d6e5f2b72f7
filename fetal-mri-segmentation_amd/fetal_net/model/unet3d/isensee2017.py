"""`isensee2017_model_3d` with the reference signature and topology (reference fetal_net/model/unet3d/isensee2017.py:15-111),
executed by the generic layer-graph engine (fmri_hip.graph_engine).

Per level: in-conv block (3x3x3, stride 2 below the first level) + context module (block -> SpatialDropout3D -> block), summed;
decoder: UpSampling3D -> block, concatenate([skip, up]), localisation (3x3x3 block -> 1x1x1 block); 1x1x1 segmentation heads on the
`n_segmentation_levels` shallowest levels, summed bottom-up through UpSampling3D; Activation(activation_name).  Every block is
Conv3D -> keras-contrib InstanceNormalization(axis=1) -> LeakyReLU.
"""
from ...engine_model import Adam, Model
from ...metrics import dice_coefficient, dice_coefficient_loss, vod_coefficient
from ..graph import Graph


def _block(g, x, n_filters, kernel=(3, 3, 3), strides=(1, 1, 1)):
    h = g.conv(x, n_filters, kernel, strides=strides, padding='same')
    h = g.instance_norm(h, axis=1)
    return g.leaky_relu(h)


def isensee2017_model_3d(input_shape=(1, 128, 128, 128), n_base_filters=16, depth=5, dropout_rate=0.3, n_segmentation_levels=1,
                         n_labels=1, optimizer=Adam, initial_learning_rate=5e-4, loss_function=dice_coefficient_loss,
                         activation_name="sigmoid", mask_shape=None, **kargs):
    input_shape = tuple(int(v) for v in input_shape)
    g = Graph()
    x = g.input(input_shape)
    cur = x
    level_out, level_filters = [], []
    for level in range(depth):
        n = (2 ** level) * n_base_filters
        level_filters.append(n)
        in_conv = _block(g, cur, n) if level == 0 else _block(g, cur, n, strides=(2, 2, 2))
        c = _block(g, in_conv, n)
        c = g.spatial_dropout(c, dropout_rate, 3)
        c = _block(g, c, n)
        cur = g.add([in_conv, c])
        level_out.append(cur)
    heads = {}
    for level in range(depth - 2, -1, -1):
        up = _block(g, g.up_sample(cur, (2, 2, 2)), level_filters[level])
        cat = g.concat([level_out[level], up], axis=1)
        cur = _block(g, _block(g, cat, level_filters[level]), level_filters[level], kernel=(1, 1, 1))
        if level < n_segmentation_levels:
            heads[level] = g.conv(cur, n_labels, (1, 1, 1))
    out = None
    for level in reversed(range(n_segmentation_levels)):
        out = heads[level] if out is None else g.add([out, heads[level]])
        if level > 0:
            out = g.up_sample(out, (2, 2, 2))
    g.activation(out, activation_name)
    # Keras records SpatialDropout3D(rate, data_format) — keep the attribute the callers may inspect
    for l in g.layers:
        if l.class_name == "SpatialDropout3D":
            l.config["data_format"] = "channels_first"
    builder_kwargs = dict(input_shape=input_shape, n_base_filters=n_base_filters, depth=depth, dropout_rate=dropout_rate,
                          n_segmentation_levels=n_segmentation_levels, n_labels=n_labels, initial_learning_rate=initial_learning_rate,
                          loss_function=loss_function, activation_name=activation_name)
    if "compute_dtype" in kargs:
        builder_kwargs["compute_dtype"] = kargs["compute_dtype"]
    if mask_shape is not None:
        builder_kwargs["mask_shape"] = tuple(int(v) for v in mask_shape)
    model = Model(g.layers, None, "isensee2017_model_3d", builder_kwargs, "channels_first_3d", name="isensee2017_3d_Model")
    model._graph_engine = True
    unsupported = []
    if mask_shape is not None:
        # reference isensee2017.py:85-88: a second input carries the distance mask and the loss factory closes over it
        from ...metrics import MaskInput
        loss_function = loss_function(MaskInput())
        model._mask_shape = tuple(int(v) for v in mask_shape)
        if not getattr(loss_function, "mask_weighted", False):
            unsupported.append("mask_shape with a loss factory other than dice_and_xent_mask")
    if activation_name != "sigmoid":
        unsupported.append("activation_name != 'sigmoid'")
    if unsupported:
        model._unsupported = ", ".join(unsupported)
    metrics = ['binary_accuracy', vod_coefficient]
    if loss_function != dice_coefficient_loss:
        metrics += [dice_coefficient]
    model.compile(optimizer=optimizer(lr=initial_learning_rate), loss=loss_function, metrics=metrics)
    return model
