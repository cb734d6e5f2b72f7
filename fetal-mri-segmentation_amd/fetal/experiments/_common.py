"""What train_adv.py and train_semi.py share: config defaults, model construction, generator construction
(reference fetal/experiments/train_adv.py:24-35, :127-165, :183-206)."""
import glob


def config_with_defaults(config):
    for key, value in (("dis_model_name", "discriminator_image_3d"), ("dis_loss", "binary_crossentropy_loss"), ("gen_steps", 1), ("dis_steps", 1),
                       ("gd_loss_ratio", 10)):
        config.setdefault(key, value)
    if config["dis_model_name"] == "discriminator_image":
        # the reference's default names a builder its model package never exported (fetal_net/model/__init__.py:17-18 has only the
        # _2d / _3d ones); the 3-D PatchGAN is the one its experiment scripts were written around
        config["dis_model_name"] = "discriminator_image_3d"
    return config


def build_models(config, overwrite):
    import fetal_net
    import fetal_net.metrics
    import fetal_net.model
    from fetal.utils import get_last_model_path
    seg_loss_func = getattr(fetal_net.metrics, config['loss'])
    dis_loss_func = getattr(fetal_net.metrics, config['dis_loss'])
    gen_model = getattr(fetal_net.model, config['model_name'])(
        input_shape=config["input_shape"], initial_learning_rate=config["initial_learning_rate"],
        **{'dropout_rate': config['dropout_rate'], 'loss_function': seg_loss_func,
           'mask_shape': None if config["weight_mask"] is None else config["input_shape"], 'old_model_path': config['old_model']})
    dis_model = getattr(fetal_net.model, config['dis_model_name'])(
        input_shape=[config["input_shape"][0] + config["n_labels"]] + config["input_shape"][1:],
        initial_learning_rate=config["initial_learning_rate"], **{'dropout_rate': config['dropout_rate'], 'loss_function': dis_loss_func})
    if not overwrite and len(glob.glob(config["model_file"] + 'g_*.h5')) > 0:
        gen_model_path = get_last_model_path(config["model_file"] + 'g_')
        print('Loading gen model from: {}'.format(gen_model_path))
        gen_model.load_weights(gen_model_path)
    gen_model.summary()
    dis_model.summary()
    return gen_model, dis_model


def generator_kwargs(config, overwrite, **override):
    kw = dict(batch_size=config["batch_size"], data_split=config["validation_split"], overwrite=overwrite,
              validation_keys_file=config["validation_file"], training_keys_file=config["training_file"], test_keys_file=config["test_file"],
              n_labels=config["n_labels"], labels=config["labels"], patch_shape=(*config["patch_shape"], config["patch_depth"]),
              validation_batch_size=config["validation_batch_size"], augment=config["augment"], skip_blank_train=config["skip_blank_train"],
              skip_blank_val=config["skip_blank_val"], truth_index=config["truth_index"], truth_size=config["truth_size"],
              prev_truth_index=config["prev_truth_index"], prev_truth_size=config["prev_truth_size"],
              truth_downsample=config["truth_downsample"], truth_crop=config["truth_crop"], patches_per_epoch=config["patches_per_epoch"],
              categorical=config["categorical"], is3d=config["3D"], drop_easy_patches_train=config["drop_easy_patches_train"],
              drop_easy_patches_val=config["drop_easy_patches_val"])
    kw.update(override)
    return kw
