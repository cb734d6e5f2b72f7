"""`unet_model_3d` with the reference signature and topology (reference fetal_net/model/unet3d/unet.py:17-138), returning
a Keras-Model duck type whose compute runs on the MI355X engine.

Topology: `depth` encoder levels of two [Conv3D 3x3x3 'same' -> (BatchNorm | InstanceNorm)? -> ReLU] blocks with
n_base_filters*2^level and twice that many filters, MaxPooling3D between levels; decoder levels of
(UpSampling3D | Conv3DTranspose k2 s2) -> concatenate([up, skip], axis=1) -> two conv blocks with the skip's filter count;
Conv3D(n_labels, 1x1x1) -> Activation(activation_name).  Compiled with Adam(lr) and metrics
['binary_accuracy', vod_coefficient] (+ dice_coefficient when the loss is not the Dice loss).
"""
from ...engine_model import Adam, Model
from ...metrics import dice_coefficient, dice_coefficient_loss, vod_coefficient
from ..graph import Graph


def conv_block(g, x, n_filters, batch_normalization=False, kernel=(3, 3, 3), activation=None, padding='same', strides=(1, 1, 1),
               instance_normalization=False):
    """one [conv -> norm? -> activation] block (reference create_convolution_block, unet.py:89-115)"""
    h = g.conv(x, n_filters, kernel, strides=strides, padding=padding)
    if batch_normalization:
        h = g.batch_norm(h, axis=1)
    elif instance_normalization:
        h = g.instance_norm(h, axis=1)
    if activation is None:
        return g.activation(h, 'relu')
    if activation == 'leaky_relu':
        return g.leaky_relu(h)
    return g.activation(h, activation)


def up_block(g, x, pool_size, deconvolution, kernel_size=(2, 2, 2), strides=(2, 2, 2)):
    """reference get_up_convolution (unet.py:132-138): transposed conv keeps the channel count, else nearest x2"""
    if deconvolution:
        return g.deconv(x, x.shape[1], kernel_size, strides)
    return g.up_sample(x, pool_size)


def unet_model_3d(input_shape, pool_size=(2, 2, 2), n_labels=1, initial_learning_rate=0.00001, deconvolution=False, depth=4,
                  n_base_filters=32, include_label_wise_dice_coefficients=False, batch_normalization=False,
                  activation_name="sigmoid", loss_function=dice_coefficient_loss, **kargs):
    input_shape = tuple(int(v) for v in input_shape)
    pool_size = tuple(pool_size)
    g = Graph()
    x = g.input(input_shape)
    skips = []
    h = x
    for level in range(depth):
        h = conv_block(g, h, n_base_filters * (2 ** level), batch_normalization=batch_normalization)
        h = conv_block(g, h, n_base_filters * (2 ** level) * 2, batch_normalization=batch_normalization)
        skips.append(h)
        if level < depth - 1:
            h = g.max_pool(h, pool_size)
    for level in range(depth - 2, -1, -1):
        up = up_block(g, h, pool_size, deconvolution)
        cat = g.concat([up, skips[level]], axis=1)
        h = conv_block(g, cat, skips[level].shape[1], batch_normalization=batch_normalization)
        h = conv_block(g, h, skips[level].shape[1], batch_normalization=batch_normalization)
    h = g.conv(h, n_labels, (1, 1, 1))
    g.activation(h, activation_name)

    unsupported = []
    if pool_size != (2, 2, 2):
        unsupported.append("pool_size != (2,2,2)")
    if activation_name != "sigmoid":
        unsupported.append("activation_name != 'sigmoid'")
    builder_kwargs = dict(input_shape=input_shape, pool_size=pool_size, n_labels=n_labels, initial_learning_rate=initial_learning_rate,
                          deconvolution=deconvolution, depth=depth, n_base_filters=n_base_filters,
                          batch_normalization=batch_normalization, activation_name=activation_name, loss_function=loss_function)
    if "compute_dtype" in kargs:
        builder_kwargs["compute_dtype"] = kargs["compute_dtype"]
    plan_args = dict(in_channels=input_shape[0], spatial=input_shape[1:], depth=depth, n_base_filters=n_base_filters,
                     n_labels=n_labels, ndim=3, norm="batch" if batch_normalization else None, deconvolution=bool(deconvolution))
    model = Model(g.layers, plan_args, "unet_model_3d", builder_kwargs, "channels_first_3d")
    if unsupported:
        model._unsupported = ", ".join(unsupported)
    metrics = ['binary_accuracy', vod_coefficient]
    if loss_function != dice_coefficient_loss:
        metrics += [dice_coefficient]
    model.compile(optimizer=Adam(lr=initial_learning_rate), loss=loss_function, metrics=metrics)
    return model


# reference-named helpers (other builders import them from here: reference isensee2017.py:7)
def create_convolution_block(input_layer, n_filters, batch_normalization=False, kernel=(3, 3, 3), activation=None, padding='same',
                             strides=(1, 1, 1), instance_normalization=False, graph=None):
    if graph is None:
        raise TypeError("create_convolution_block needs the recording graph (graph=...) in this implementation")
    return conv_block(graph, input_layer, n_filters, batch_normalization, kernel, activation, padding, strides, instance_normalization)


def compute_level_output_shape(n_filters, depth, pool_size, image_shape):
    return tuple([None, n_filters] + [int(s // (p ** depth)) for s, p in zip(image_shape, pool_size)])
