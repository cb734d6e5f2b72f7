"""The reference's generator module under its name (fetal_net/generator.py), over the device generator.

`get_training_and_validation_generators` is what the reference's training scripts call (fetal/train_fetal.py:46-72): it makes / reloads the
train / validation / test split (get_validation_split, pickled index lists), counts the steps of an epoch and returns the two endless batch
generators.  Same keywords, same split files, same step counts; the generators are `fetal_net.device_generator.device_data_generator` over ONE
copy of the padded volumes in HBM (DeviceDataFile: the reference's DataFileDummy + pad_samples, generator.py:13-57), so the batches are CUDA tensors
that `train_model` / `fit_generator` consume without a host round trip.  Not carried over: `truth_downsample` > 1 (NotImplementedError in the
device generator) and `truth_crop`, which the reference's own data_generator accepts and ignores.

The small pure helpers (get_number_of_steps, get_multi_class_labels, split_list, the two index generators) are restated as they are."""
import numpy as np

from .data import get_validation_split, split_list  # noqa: F401  (reference generator.py:158-191 keeps both here)
from .device_generator import DeviceDataFile, device_data_generator, list_generator, random_list_generator  # noqa: F401

data_generator = device_data_generator                     # reference generator.py:222 (same keywords; yields device tensors)


def get_number_of_steps(n_samples, batch_size):
    """reference generator.py:149-155"""
    if n_samples <= batch_size:
        return n_samples
    if np.remainder(n_samples, batch_size) == 0:
        return n_samples // batch_size
    return n_samples // batch_size + 1


def get_multi_class_labels(data, n_labels, labels=None):
    """label map (n_samples, 1, ...) -> binary int8 (n_samples, n_labels, ...): channel k marks label labels[k] (k + 1 without `labels`)
    (reference generator.py:404-419)"""
    data = np.asarray(data)
    y = np.zeros([data.shape[0], n_labels] + list(data.shape[2:]), np.int8)
    for k in range(n_labels):
        y[:, k][data[:, 0] == (labels[k] if labels is not None else k + 1)] = 1
    return y


def get_training_and_validation_generators(data_file, batch_size, n_labels, training_keys_file, validation_keys_file, test_keys_file,
                                           patch_shape=None, data_split=0.8, overwrite=False, labels=None, augment=None,
                                           validation_batch_size=None, skip_blank_train=True, skip_blank_val=False, truth_index=-1, truth_size=1,
                                           truth_downsample=None, truth_crop=True, patches_per_epoch=1, categorical=True, is3d=False,
                                           prev_truth_index=None, prev_truth_size=None, drop_easy_patches_train=False,
                                           drop_easy_patches_val=False, samples_pad=3, val_augment=None, device="cuda", verbose=True):
    """reference generator.py:58-146 -> (training generator, validation generator, training steps, validation steps).  `data_file`: an opened
    data file (fetal_net.data.open_data_file, or anything with .root.data / .root.truth [/ .root.mask, .root.subject_ids])."""
    if not validation_batch_size:
        validation_batch_size = batch_size
    training_list, validation_list, test_list = get_validation_split(data_file, data_split=data_split, overwrite=overwrite,
                                                                     training_file=training_keys_file, validation_file=validation_keys_file,
                                                                     test_file=test_keys_file)
    if verbose and hasattr(data_file.root, "subject_ids"):
        ids = data_file.root.subject_ids
        for tag, lst in (("Training", training_list), ("Validation", validation_list), ("Test", test_list)):
            print("{}: {}".format(tag, [ids[i].decode() if isinstance(ids[i], bytes) else str(ids[i]) for i in lst]))
    # the number of steps of an epoch as the reference sets it: patches_per_epoch over the batch size (generator.py:117-122)
    num_training_steps = patches_per_epoch // batch_size
    num_validation_steps = patches_per_epoch // validation_batch_size
    if verbose:
        print("Number of training steps: ", num_training_steps)
        print("Number of validation steps: ", num_validation_steps)
    # one resident copy of every volume either generator samples from
    ddf = DeviceDataFile(data_file, patch_shape, samples_pad, truth_downsample, indices=sorted(set(training_list) | set(validation_list)),
                         device=device)
    common = dict(n_labels=n_labels, labels=labels, patch_shape=patch_shape, truth_index=truth_index, truth_size=truth_size,
                  truth_downsample=truth_downsample, truth_crop=truth_crop, categorical=categorical, is3d=is3d, prev_truth_index=prev_truth_index,
                  prev_truth_size=prev_truth_size, samples_pad=samples_pad, device=device)
    training_generator = device_data_generator(ddf, training_list, batch_size=batch_size, augment=augment, skip_blank=skip_blank_train,
                                               drop_easy_patches=drop_easy_patches_train, **common)
    validation_generator = device_data_generator(ddf, validation_list, batch_size=validation_batch_size, augment=val_augment,
                                                 skip_blank=skip_blank_val, drop_easy_patches=drop_easy_patches_val, noise_seed=1, **common)
    return training_generator, validation_generator, num_training_steps, num_validation_steps
