#!/usr/bin/env python3
"""Rewrite a reference PyTables data file (fetal_net/data.py:11-17: VLArrays of pickled arrays, blosc level 5) into the plain HDF5 layout
`fetal_net.data.open_data_file` reads without PyTables or the blosc plug-in.

Runs in the REFERENCE's environment (needs `tables` + numpy only, nothing of this package) - and, where PyTables is absent, through this
package's own reader of the PyTables layout (fetal_net.data.PyTablesDataFile):

    python tools/convert_data_file.py fetal_data.h5 fetal_data_plain.h5 [--float32]

Layout written (plain contiguous datasets, no filters): root attributes fmri_data_file = 1, n_samples; /data/s<i>, /truth/s<i>,
/mask/s<i> (when the source has masks), /subject_ids (when present).  --float32 stores the volumes as float32 (half the size; the
device path converts to float32 anyway), default keeps the reference's float64.
"""
import argparse
import sys
import warnings

import numpy as np


def convert_without_pytables(a):
    """no PyTables here: read the source with this package's own reader (libhdf5 + its blosc decoder) and write the plain layout"""
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fetal-mri-segmentation_amd"))
    from fetal_net.data import PyTablesDataFile, write_plain_data_file
    with PyTablesDataFile(a.src) as src:
        n = len(src.root.data)
        data = [np.asarray(src.root.data[i]) for i in range(n)]
        if a.float32:
            data = [d.astype(np.float32) for d in data]
        truth = [np.asarray(src.root.truth[i]) for i in range(n)]
        mask = [np.asarray(src.root.mask[i]) for i in range(n)] if "mask" in src.root and len(src.root.mask) == n else None
        ids = src.root.subject_ids if "subject_ids" in src.root else None
        write_plain_data_file(a.dst, data, truth, mask, ids)
    print("wrote %s: %d samples (without PyTables)" % (a.dst, n))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--float32", action="store_true")
    a = ap.parse_args()
    try:
        import tables
    except ImportError:
        return convert_without_pytables(a)
    warnings.simplefilter("ignore", tables.NaturalNameWarning)
    with tables.open_file(a.src, "r") as src, tables.open_file(a.dst, "w") as dst:
        n = len(src.root.data)
        dst.root._v_attrs.fmri_data_file = np.int32(1)
        dst.root._v_attrs.n_samples = np.int32(n)
        for name in ("data", "truth", "mask"):
            if name not in src.root or len(getattr(src.root, name)) == 0:
                continue
            g = dst.create_group(dst.root, name)
            for i in range(n):
                arr = np.asarray(getattr(src.root, name)[i])
                if name == "data" and a.float32:
                    arr = arr.astype(np.float32)
                dst.create_array(g, "s%d" % i, obj=np.ascontiguousarray(arr))
        if "subject_ids" in src.root:
            dst.create_array(dst.root, "subject_ids", obj=np.asarray([bytes(s) for s in src.root.subject_ids[:]], dtype="S"))
    print("wrote %s: %d samples" % (a.dst, n))


if __name__ == "__main__":
    sys.exit(main())
