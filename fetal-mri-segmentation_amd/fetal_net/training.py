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


def step_decay(epoch, initial_lrate, drop, epochs_drop):
    return initial_lrate * math.pow(drop, math.floor((1 + epoch) / float(epochs_drop)))


def get_callbacks(model_file, initial_learning_rate=0.0001, learning_rate_drop=0.5, learning_rate_epochs=None,
                  learning_rate_patience=50, logging_file="training.log", verbosity=1, early_stopping_patience=None):
    """[checkpoint, csv log, lr policy, (early stopping)] in that order (reference test/test_training.py:10-15)."""
    callbacks = [ModelCheckpoint(model_file + '-epoch{epoch:02d}-loss{val_loss:.3f}-acc{val_binary_accuracy:.3f}.h5',
                                 save_best_only=True, verbose=verbosity, monitor='val_loss'),
                 CSVLogger(logging_file, append=True)]
    if learning_rate_epochs:
        callbacks.append(LearningRateScheduler(partial(step_decay, initial_lrate=initial_learning_rate, drop=learning_rate_drop,
                                                       epochs_drop=learning_rate_epochs)))
    else:
        callbacks.append(ReduceLROnPlateau(factor=learning_rate_drop, patience=learning_rate_patience, verbose=verbosity))
    if early_stopping_patience:
        callbacks.append(EarlyStopping(verbose=verbosity, patience=early_stopping_patience))
    return callbacks


def load_old_model(model_file, verbose=True, config=None) -> Model:
    """Re-open a checkpoint written by `Model.save` (the file records its builder + arguments); with `config` given and an
    unreadable header, rebuild from the config and load the weights (reference training.py:66-86)."""
    print("Loading pre-trained model")
    if verbose:
        print('Loading model from {}...'.format(model_file))
    try:
        meta = read_checkpoint_meta(model_file)
        kwargs = dict(meta["builder_kwargs"])
        for k, v in list(kwargs.items()):
            if isinstance(v, dict) and "__callable__" in v:
                kwargs[k] = getattr(fetal_net.metrics, v["__callable__"])
        model = getattr(fetal_net.model, meta["builder"])(**kwargs)
        model.load_weights(model_file)
        if meta.get("optimizer"):
            model.optimizer.lr = float(meta["optimizer"]["lr"])
        return model
    except (ValueError, KeyError, OSError) as error:
        print(error)
        if config is None:
            raise
        print('Trying to build model manually...')
        loss_func = getattr(fetal_net.metrics, config['loss'])
        model_func = getattr(fetal_net.model, config['model_name'])
        model = model_func(input_shape=config["input_shape"], initial_learning_rate=config["initial_learning_rate"],
                           **{'dropout_rate': config['dropout_rate'], 'loss_function': loss_func,
                              'mask_shape': None if config["weight_mask"] is None else config["input_shape"],
                              'old_model_path': config['old_model']})
        model.load_weights(model_file)
        return model


def train_model(model, model_file, training_generator, validation_generator, steps_per_epoch, validation_steps,
                initial_learning_rate=0.001, learning_rate_drop=0.5, learning_rate_epochs=None, n_epochs=500,
                learning_rate_patience=20, early_stopping_patience=None, output_folder='.'):
    return model.fit_generator(generator=training_generator, steps_per_epoch=steps_per_epoch, epochs=n_epochs,
                               validation_data=validation_generator, validation_steps=validation_steps, max_queue_size=15,
                               workers=1, use_multiprocessing=False,
                               callbacks=get_callbacks(model_file, initial_learning_rate=initial_learning_rate,
                                                       learning_rate_drop=learning_rate_drop,
                                                       learning_rate_epochs=learning_rate_epochs,
                                                       learning_rate_patience=learning_rate_patience,
                                                       early_stopping_patience=early_stopping_patience,
                                                       logging_file=os.path.join(output_folder, 'training')))
