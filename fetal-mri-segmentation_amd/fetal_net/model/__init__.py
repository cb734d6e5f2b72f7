"""Builders looked up by name: `getattr(fetal_net.model, config['model_name'])` (reference fetal/train_fetal.py:32,
fetal_net/model/__init__.py:3-18).  The hot-path builders of SURVEY.md §8a: both U-Nets and both Isensee networks;
§8f row 4: the PatchGAN discriminators of the adversarial experiments."""
from .unet3d.unet import unet_model_3d
from .unet3d.isensee2017 import isensee2017_model_3d
from .unet.unet import unet_model_2d
from .unet.isensee import isensee2017_model
from .discriminator.all_dis_2d import discriminator_image_2d
from .discriminator.all_dis_3d import discriminator_image_3d
