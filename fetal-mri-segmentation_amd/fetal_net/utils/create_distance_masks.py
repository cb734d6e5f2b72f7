"""Distance masks for the mask-weighted loss (reference fetal_net/utils/create_distance_masks.py, a script: for every truth volume the Euclidean
distance of each voxel to the label's border, inside and outside, in millimetres; stored as the `mask` entry of the data file and sampled by the
generators beside the truth).  Here as a function plus the script's loop."""
import glob
import os

import numpy as np
from scipy import ndimage


def distance_mask(mask, sampling=(0.4, 0.4, 3.0)):
    """distance_transform_edt(mask) + distance_transform_edt(1 - mask) with the voxel spacing `sampling` (reference :17-19)"""
    mask = np.asarray(mask)
    return ndimage.distance_transform_edt(mask, sampling=sampling) + ndimage.distance_transform_edt(1 - mask, sampling=sampling)


def create_distance_masks(dataset_folder, ext=".gz", sampling=(0.4, 0.4, 3.0)):
    """<dataset_folder>/*/truth.nii<ext> -> dists.nii.gz beside each (identity affine, as the reference writes it) -> the written paths"""
    from .nifti import load_nifti, save_nifti
    out = []
    for mask_path in sorted(glob.glob(os.path.join(dataset_folder, "*", "truth.nii" + ext))):
        dists = distance_mask(load_nifti(mask_path), sampling)
        out.append(save_nifti(dists, os.path.join(os.path.dirname(mask_path), "dists.nii.gz"), np.eye(4)))
    return out


if __name__ == "__main__":
    import sys
    for path in create_distance_masks(sys.argv[1] if len(sys.argv) > 1 else ""):
        print(path)
