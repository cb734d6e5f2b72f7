"""Drop-in replacements of the reference's experiment scripts (fetal/experiments/train_adv.py, train_semi.py): the reference files
drive Keras directly (keras.Input / Model / Network / K.set_value), so they cannot run over the engine-backed models unchanged; these
keep the command line, the config keys and the loop, and hand the training to fetal_net.adversarial."""
