"""`python -m fetal.experiments.train_semi --config_dir <dir>`: semi-supervised variant (reference fetal/experiments/train_semi.py:126-329):
the segmentation loss comes from the labelled stream, the adversarial term from a second, unlabelled stream (the validation split's
generator with the training augmentation, train_semi.py:215-240)."""
import os


def main(overwrite=False, config=None):
    from fetal.utils import create_data_file
    from fetal_net.adversarial import train_adversarial
    from fetal_net.data import open_data_file
    from fetal_net.generator import get_training_and_validation_generators
    from ._common import build_models, config_with_defaults, generator_kwargs
    if config is None:
        from fetal.config_utils import get_config
        config = get_config()
    config = config_with_defaults(config)
    if overwrite or not os.path.exists(config["data_file"]):
        create_data_file(config)
    data_file_opened = open_data_file(config["data_file"])
    gen_model, dis_model = build_models(config, overwrite)
    train_generator, validation_generator, n_train_steps, n_validation_steps = get_training_and_validation_generators(
        data_file_opened, **generator_kwargs(config, overwrite))
    semi_kw = generator_kwargs(config, overwrite, val_augment=config["augment"])
    semi_kw.pop("augment")
    _, semi_generator, _, _ = get_training_and_validation_generators(data_file_opened, **semi_kw)
    try:
        return train_adversarial(config, gen_model, dis_model, train_generator, validation_generator, n_train_steps, n_validation_steps,
                                 semi_generator=semi_generator)
    finally:
        data_file_opened.close()


if __name__ == "__main__":
    from fetal.config_utils import get_config
    cfg = get_config()
    main(overwrite=cfg["overwrite"], config=cfg)
