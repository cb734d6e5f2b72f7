"""`isensee2017_model` - the 2-D Isensee network - with the reference signature and topology (reference
fetal_net/model/unet/isensee.py:14-105), executed by the generic layer-graph engine (fmri_hip.graph_engine) in planar mode.

Channels-LAST input (X, Y, C) between two Permute layers, like `unet_model_2d`.  Per level: in-conv block (3x3, stride 2 below the first
level) + context module (block -> SpatialDropout2D -> block), summed; decoder: UpSampling2D -> block, concatenate([skip, up]),
localisation (3x3 block -> 1x1 block); a 1x1 segmentation head on each of the `n_segmentation_levels` shallowest levels.  Unlike the 3-D
builder the heads are only summed (bottom-up through UpSampling2D) when `summation=True`; by default the output is the head of level 0
alone and the deeper heads - although the reference creates them, which advances Keras' layer-name counters - are not part of the model.
Every block is Conv2D -> keras-contrib InstanceNormalization(axis=1) -> LeakyReLU.
"""
from ...engine_model import Adam, Model
from ...metrics import dice_coefficient, dice_coefficient_loss, vod_coefficient
from ..graph import Graph


def _block(g, x, n_filters, kernel=(3, 3), strides=(1, 1)):
    h = g.conv(x, n_filters, kernel, strides=strides, padding='same')
    h = g.instance_norm(h, axis=1)
    return g.leaky_relu(h)


def _reachable(layers):
    """the layers a Keras Model(inputs, outputs) would keep: those the output depends on, in creation order"""
    by_name = dict((l.name, l) for l in layers)
    keep, stack = set(), [layers[-1].name]
    while stack:
        n = stack.pop()
        if n not in keep:
            keep.add(n)
            stack.extend(by_name[n].inbound)
    return [l for l in layers if l.name in keep]


def isensee2017_model(input_shape=(128, 128, 5), n_base_filters=16, depth=5, dropout_rate=0.3, n_segmentation_levels=3, n_labels=1,
                      optimizer=Adam, initial_learning_rate=5e-4, loss_function=dice_coefficient_loss, activation_name="sigmoid",
                      summation=False, **kargs):
    input_shape = tuple(int(v) for v in input_shape)
    g = Graph()
    x = g.input(input_shape)
    cur = g.permute(x, (3, 1, 2))
    level_out, level_filters = [], []
    for level in range(depth):
        n = (2 ** level) * n_base_filters
        level_filters.append(n)
        in_conv = _block(g, cur, n) if level == 0 else _block(g, cur, n, strides=(2, 2))
        c = _block(g, in_conv, n)
        c = g.spatial_dropout(c, dropout_rate, 2)
        c = _block(g, c, n)
        cur = g.add([in_conv, c])
        level_out.append(cur)
    heads = {}
    for level in range(depth - 2, -1, -1):
        up = _block(g, g.up_sample(cur, (2, 2)), level_filters[level])
        cat = g.concat([level_out[level], up], axis=1)
        cur = _block(g, _block(g, cat, level_filters[level]), level_filters[level], kernel=(1, 1))
        if level < n_segmentation_levels:
            heads[level] = g.conv(cur, n_labels, (1, 1))
    if summation:
        out = None
        for level in reversed(range(n_segmentation_levels)):
            out = heads[level] if out is None else g.add([out, heads[level]])
            if level > 0:
                out = g.up_sample(out, (2, 2))
    else:
        out = heads[0]
    g.permute(g.activation(out, activation_name), (2, 3, 1))
    for l in g.layers:
        if l.class_name == "SpatialDropout2D":
            l.config["data_format"] = "channels_first"
    builder_kwargs = dict(input_shape=input_shape, n_base_filters=n_base_filters, depth=depth, dropout_rate=dropout_rate,
                          n_segmentation_levels=n_segmentation_levels, n_labels=n_labels, initial_learning_rate=initial_learning_rate,
                          loss_function=loss_function, activation_name=activation_name, summation=summation)
    if "compute_dtype" in kargs:
        builder_kwargs["compute_dtype"] = kargs["compute_dtype"]
    model = Model(_reachable(g.layers), None, "isensee2017_model", builder_kwargs, "channels_last_2d", name="isensee2017_2d_Model")
    model._graph_engine = True
    model._created_layers = list(g.layers)       # incl. the heads Keras would drop: what the reference builder CREATED (name counters)
    if activation_name != "sigmoid":
        model._unsupported = "activation_name != 'sigmoid'"
    metrics = ['binary_accuracy', vod_coefficient]
    if loss_function != dice_coefficient_loss:
        metrics += [dice_coefficient]
    model.compile(optimizer=optimizer(lr=initial_learning_rate), loss=loss_function, metrics=metrics)
    return model
