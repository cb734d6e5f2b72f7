"""Data-file interchange (reference fetal_net/data.py:11-78, fetal/utils.py:11-39, fetal_net/generator.py:158-190).

The reference keeps its volumes in a PyTables file: three VLArrays of pickled numpy arrays (`/data`, `/truth`, `/mask`, blosc level 5)
plus `/subject_ids`, and everything downstream only uses `data_file.root.<name>[i]`, `len(data_file.root.data)`,
`'subject_ids' in data_file.root` and `.close()`.  `open_data_file` returns an object with exactly that surface for

  * a PyTables file as the reference writes it: through `tables` when that is installed (read-write), otherwise read-only through
    `PyTablesDataFile` - libhdf5 over ctypes, the blosc chunk filter decoded by fetal_net/utils/blosc.py, the rows unpickled; pinned
    by a file written with the reference's own functions under PyTables 3.6.1 (tests/golden/pytables_data_golden.h5), and
  * the PLAIN layout below, through the same binding - no filter at all:
        /            attribute fmri_data_file = 1, n_samples
        /data/s<i>   one dataset per sample (any float dtype; the reference stores float64), /truth/s<i> (uint8), /mask/s<i> (optional)
        /subject_ids fixed-length byte strings (optional)
    written by `write_plain_data_file` here or by `tools/convert_data_file.py`, which runs where PyTables exists and rewrites a
    reference data file sample by sample (PyTables itself writes the plain arrays, so the converter needs nothing of this package).

Also here: the train / validation / test split pickles and `norm_params.json` exactly as the reference writes them.
"""
import json
import os
import pickle
import random

import numpy as np

PLAIN_MARK = "fmri_data_file"


class _Samples(object):
    """`root.data`-like sequence over the datasets s0, s1, ... of one group"""

    def __init__(self, group, n, dtype=None):
        self._g, self._n, self._dtype = group, n, dtype

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        i = int(i)
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        d = self._g["s%d" % i]
        a = np.asarray(d[()])
        d.close()
        return a

    def __iter__(self):
        return (self[i] for i in range(self._n))


class _Root(object):
    def __init__(self, names):
        self._names = set(names)

    def __contains__(self, name):
        return name in self._names


class PlainDataFile(object):
    """read-only view of a plain-layout data file with the PyTables surface the reference uses"""

    def __init__(self, filename):
        from .utils import hdf5
        self.filename = filename
        self._f = hdf5.File(filename, "r")
        if PLAIN_MARK not in self._f.attrs:
            self._f.close()
            raise ValueError("%s is not a plain data file (no %s attribute)" % (filename, PLAIN_MARK))
        n = int(np.asarray(self._f.attrs["n_samples"]).reshape(-1)[0])
        names = [k for k in ("data", "truth", "mask") if k in self._f]
        self.root = _Root(names + (["subject_ids"] if "subject_ids" in self._f else []))
        self._groups = []
        for k in names:
            g = self._f[k]
            self._groups.append(g)
            setattr(self.root, k, _Samples(g, n))
        if "subject_ids" in self._f:
            d = self._f["subject_ids"]
            self.root.subject_ids = [bytes(v) for v in np.atleast_1d(np.asarray(d[()]))]
            d.close()

    def close(self):
        for g in self._groups:
            g.close()
        self._groups = []
        if self._f is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


class _Pickled(object):
    """`root.data`-like sequence over a VLArray whose rows are pickles (PyTables ObjectAtom)"""

    def __init__(self, rows):
        self._rows = rows

    def __len__(self):
        return len(self._rows)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        import warnings
        with warnings.catch_warnings():                      # pickles of older numpy versions name numpy.core.*: still loadable, noisily
            warnings.simplefilter("ignore", DeprecationWarning)
            return pickle.loads(self._rows[i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class PyTablesDataFile(object):
    """read-only view of a reference data file (reference data.py:11-17, :33-38, :65-66) without PyTables: `/data`, `/truth`, `/mask` are
    1-D datasets of variable-length bytes (one pickled array per subject) whose chunks carry the blosc filter, `/subject_ids` a plain
    string array.  An empty `/mask` (data.py:36-38 appends masks only when the subject has one) reads as a zero-length sequence."""

    def __init__(self, filename):
        from .utils import hdf5
        self.filename = filename
        self._f = hdf5.File(filename, "r")
        self._sets = []
        names = []
        try:
            for k in ("data", "truth", "mask"):
                if k in self._f:
                    d = self._f[k]
                    if not isinstance(d, hdf5.VLenBytes):
                        d.close()
                        raise ValueError("%s:/%s is not a VLArray of pickled rows" % (filename, k))
                    self._sets.append(d)
                    names.append(k)
            if "data" not in names or "truth" not in names:
                raise ValueError("%s has no /data and /truth VLArrays: not a reference data file" % filename)
        except Exception:
            self.close()
            raise
        self.root = _Root(names + (["subject_ids"] if "subject_ids" in self._f else []))
        for k, d in zip(names, self._sets):
            setattr(self.root, k, _Pickled(d))
        if "subject_ids" in self._f:
            d = self._f["subject_ids"]
            self.root.subject_ids = [bytes(v) for v in np.atleast_1d(np.asarray(d[()]))]
            d.close()

    def close(self):
        for d in self._sets:
            d.close()
        self._sets = []
        if self._f is not None:
            self._f.close()
            self._f = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


def is_plain_data_file(filename):
    from .utils import hdf5
    if not hdf5.is_hdf5(filename):
        return False
    with hdf5.File(filename, "r") as f:
        return PLAIN_MARK in f.attrs


def open_data_file(filename, readwrite="r"):
    """reference data.py:77-78.  Plain-layout files open through libhdf5; a PyTables file goes to PyTables where that is installed and
    to the built-in read-only reader where it is not."""
    if is_plain_data_file(filename):
        if readwrite != "r":
            raise ValueError("plain data files are read-only here; rewrite them with write_plain_data_file")
        return PlainDataFile(filename)
    try:
        import tables
    except ImportError:
        if readwrite != "r":
            raise ImportError("%s: writing a PyTables data file needs PyTables (reading does not)" % filename)
        return PyTablesDataFile(filename)
    return tables.open_file(filename, readwrite)


def write_plain_data_file(out_file, data, truth, mask=None, subject_ids=None):
    """data / truth / mask: sequences of arrays (one per subject, shapes may differ).  Dtypes are kept (the reference stores the
    volumes as float64 and the labels as uint8, data.py:34-39)."""
    from .utils import hdf5
    n = len(data)
    if len(truth) != n or (mask is not None and len(mask) not in (0, n)) or (subject_ids is not None and len(subject_ids) != n):
        raise ValueError("data, truth, mask and subject_ids must have one entry per subject")
    tmp = out_file + ".tmp"
    with hdf5.File(tmp, "w") as f:
        f.attrs[PLAIN_MARK] = np.int32(1)
        f.attrs["n_samples"] = np.int32(n)
        for name, seq in (("data", data), ("truth", truth), ("mask", mask if mask else None)):
            if seq is None:
                continue
            g = f.create_group(name)
            for i, a in enumerate(seq):
                g.create_dataset("s%d" % i, data=np.ascontiguousarray(a)).close()
            g.close()
        if subject_ids is not None:
            ids = np.asarray([s if isinstance(s, bytes) else str(s).encode("utf8") for s in subject_ids], dtype="S")
            f.create_dataset("subject_ids", data=ids).close()
    os.replace(tmp, out_file)
    return out_file


# ------------------------------------------------------------------------------------------------ split lists, normalisation record
def pickle_dump(item, out_file):
    with open(out_file, "wb") as f:
        pickle.dump(item, f)


def pickle_load(in_file):
    with open(in_file, "rb") as f:
        return pickle.load(f)


def split_list(input_list, split=0.8, shuffle_list=True):
    """reference generator.py:185-190 (one `random.shuffle`, then a cut at int(len * split))"""
    if shuffle_list:
        random.shuffle(input_list)
    cut = int(len(input_list) * split)
    return input_list[:cut], input_list[cut:]


def get_validation_split(data_file, training_file, validation_file, test_file, data_split=0.8, overwrite=False):
    """training / validation / test index lists as the reference draws and stores them (generator.py:158-182): shuffle all indices,
    the last one becomes the single test case, the rest is shuffled again and cut at `data_split`; three pickles.  Existing pickles are
    re-used unless `overwrite`.  The `random` module's global state is consumed in the reference's order, so a seeded run yields the
    reference's split."""
    if overwrite or not os.path.exists(training_file):
        print("Creating validation split...")
        samples = list(range(len(data_file.root.data)))
        random.shuffle(samples)
        test_list = [samples.pop()]
        training_list, validation_list = split_list(samples, split=data_split)
        for items, path in ((training_list, training_file), (validation_list, validation_file), (test_list, test_file)):
            pickle_dump(items, path)
        return training_list, validation_list, test_list
    print("Loading previous validation split...")
    return pickle_load(training_file), pickle_load(validation_file), pickle_load(test_file)


def save_norm_params(base_dir, mean, std):
    """`norm_params.json` next to the data file, as reference fetal/utils.py:21-22 writes it"""
    def plain(v):
        return None if v is None else (v.tolist() if hasattr(v, "tolist") else v)
    path = os.path.join(base_dir, "norm_params.json")
    with open(path, mode="w") as f:
        json.dump({"mean": plain(mean), "std": plain(std)}, f)
    return path


def load_norm_params(base_dir):
    with open(os.path.join(base_dir, "norm_params.json")) as f:
        d = json.load(f)
    return d.get("mean"), d.get("std")
