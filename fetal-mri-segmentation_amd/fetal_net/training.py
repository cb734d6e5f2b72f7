"""Training driver with the reference surface (reference fetal_net/training.py:22-124): `step_decay`, `get_callbacks`,
`load_old_model`, `train_model`.  The callbacks are this package's own classes (engine_model.py) with Keras 2.2 semantics;
`train_model` hands the generators to `model.fit_generator`, whose steps run on the MI355X engine."""
import math
import os
from functools import partial

import fetal_net.model
import fetal_net.metrics
from .engine_model import (CSVLogger, EarlyStopping, LearningRateScheduler, Model, ModelCheckpoint, ReduceLROnPlateau,
                           read_checkpoint_meta)


CHECKPOINT_SUFFIX = '-epoch{epoch:02d}-loss{val_loss:.3f}-acc{val_binary_accuracy:.3f}.h5'      # name pattern `get_last_model_path` globs for


def step_decay(epoch, initial_lrate, drop, epochs_drop):
    """lr after `epoch` (0-based): one multiplication by `drop` every `epochs_drop` epochs, the first after epochs_drop - 1"""
    n_drops = (epoch + 1) // float(epochs_drop)
    return initial_lrate * drop ** math.floor(n_drops)


def get_callbacks(model_file, initial_learning_rate=0.0001, learning_rate_drop=0.5, learning_rate_epochs=None,
                  learning_rate_patience=50, logging_file="training.log", verbosity=1, early_stopping_patience=None):
    """[checkpoint, csv log, lr policy, (early stopping)] in that order (reference training.py:26-42, test/test_training.py:10-15):
    best-val_loss checkpoints, an appended CSV log, a fixed step schedule when `learning_rate_epochs` is given and
    reduce-on-plateau otherwise."""
    if learning_rate_epochs:
        schedule = partial(step_decay, initial_lrate=initial_learning_rate, drop=learning_rate_drop, epochs_drop=learning_rate_epochs)
        lr_policy = LearningRateScheduler(schedule)
    else:
        lr_policy = ReduceLROnPlateau(factor=learning_rate_drop, patience=learning_rate_patience, verbose=verbosity)
    out = [ModelCheckpoint(model_file + CHECKPOINT_SUFFIX, monitor='val_loss', save_best_only=True, verbose=verbosity),
           CSVLogger(logging_file, append=True), lr_policy]
    if early_stopping_patience:
        out.append(EarlyStopping(patience=early_stopping_patience, verbose=verbosity))
    return out


def _builder_call(name, kwargs):
    """resolve loss callables recorded by name and call the builder `fetal_net.model.<name>`"""
    resolved = {}
    for k, v in kwargs.items():
        resolved[k] = getattr(fetal_net.metrics, v["__callable__"]) if (isinstance(v, dict) and "__callable__" in v) else v
    return getattr(fetal_net.model, name)(**resolved)


def _model_from_config(config):
    """the builder call of fetal/train_fetal.py:31-39 restated from a run's config.json"""
    return _builder_call(config['model_name'], dict(
        input_shape=config["input_shape"], initial_learning_rate=config["initial_learning_rate"], dropout_rate=config['dropout_rate'],
        loss_function=getattr(fetal_net.metrics, config['loss']), old_model_path=config['old_model'],
        mask_shape=config["input_shape"] if config["weight_mask"] is not None else None))


def load_old_model(model_file, verbose=True, config=None) -> Model:
    """Re-open a checkpoint: Keras-2.2 HDF5 files (the reference's ModelCheckpoint output, or `Model.save` of this package) are rebuilt
    from their `model_config`; when the header cannot be used and the run's `config` is given, the model is rebuilt from the config and
    only the weights are loaded (the fallback of reference training.py:66-86); otherwise the error propagates."""
    print("Loading pre-trained model")
    if verbose:
        print('Loading model from {}...'.format(model_file))
    try:
        meta = read_checkpoint_meta(model_file)
        model = _builder_call(meta["builder"], meta["builder_kwargs"])
        model.load_weights(model_file)
        if meta.get("optimizer"):
            model.optimizer.lr = float(meta["optimizer"]["lr"])
        return model
    except (ValueError, KeyError, OSError) as error:
        print(error)
        if config is None:
            raise
    print('Trying to build model manually...')
    model = _model_from_config(config)
    model.load_weights(model_file)
    return model


def train_model(model, model_file, training_generator, validation_generator, steps_per_epoch, validation_steps,
                initial_learning_rate=0.001, learning_rate_drop=0.5, learning_rate_epochs=None, n_epochs=500,
                learning_rate_patience=20, early_stopping_patience=None, output_folder='.'):
    """Epoch loop of reference training.py:89-124: one enqueuer thread, queue of 15 batches, callbacks of `get_callbacks`, CSV log
    `<output_folder>/training`.  Returns the History object of `fit_generator` (the reference returns None)."""
    callbacks = get_callbacks(model_file, initial_learning_rate=initial_learning_rate, learning_rate_drop=learning_rate_drop,
                              learning_rate_epochs=learning_rate_epochs, learning_rate_patience=learning_rate_patience,
                              early_stopping_patience=early_stopping_patience, logging_file=os.path.join(output_folder, 'training'))
    return model.fit_generator(generator=training_generator, steps_per_epoch=steps_per_epoch, validation_data=validation_generator,
                               validation_steps=validation_steps, epochs=n_epochs, callbacks=callbacks, max_queue_size=15, workers=1,
                               use_multiprocessing=False)
