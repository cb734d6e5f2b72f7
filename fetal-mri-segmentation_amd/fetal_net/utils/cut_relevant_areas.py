"""Bounding box of a binary mask (same names as reference fetal_net/utils/cut_relevant_areas.py:41-52); used by the two-stage
prediction to crop the region of interest the first model found."""
import numpy as np


def find_bounding_box(mask):
    """(start, end) index arrays of the smallest box holding every voxel > 0 (end exclusive)"""
    coords = np.array(np.nonzero(np.asarray(mask) > 0))
    if coords.shape[1] == 0:
        raise ValueError("empty mask: no bounding box")          # the reference fails inside np.min with a less readable message
    return coords.min(axis=1), coords.max(axis=1) + 1


def check_bounding_box(mask, start, end):
    mask = np.asarray(mask)
    return np.sum(mask[start[0]:end[0], start[1]:end[1], start[2]:end[2]]) == np.sum(mask)
