#!/opt/conda/bin/python3.9
"""Generate a GENUINE reference-format data file: PyTables VLArrays of pickled arrays under Filters(complevel=5, complib='blosc'),
written by the reference's own fetal_net/data.py functions (create_data_file / add_data_to_storage, data.py:11-38, :65-66).

Runs ONLY in the build container, and only under the stray conda interpreter that has PyTables 3.6.1 (+ its bundled c-blosc 1.20.1):
    /opt/conda/bin/python3.9 tests/golden/make_pytables_fixture.py
Outputs (committed; data, not code):
    pytables_data_golden.h5     the file as the reference writes it (3 subjects: data float64, truth uint8, mask float64; subject_ids)
    pytables_data_golden.npz    the arrays that went in (what root.data[i] / root.truth[i] / root.mask[i] must give back)
    blosc_frames_golden.npz     raw blosc frames produced by the same c-blosc for several typesizes / sizes / shuffle settings + their
                                plain contents: known-answer vectors for the blosc / blosclz decoder of fetal_net/utils/blosc.py
"""
import importlib
import importlib.util
import os
import sys
import types
import warnings

warnings.simplefilter("ignore")
import numpy as np

np.typeDict = np.sctypeDict          # PyTables 3.6.1 was built against an older numpy
np.float = float                     # reference data.py:35 uses the removed alias
import tables                         # noqa: E402

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_data_module():
    pkg = types.ModuleType("fetal_net")
    pkg.__path__ = [os.path.join(REF, "fetal_net")]
    sys.modules["fetal_net"] = pkg
    up = types.ModuleType("fetal_net.utils")
    up.__path__ = []
    sys.modules["fetal_net.utils"] = up
    uu = types.ModuleType("fetal_net.utils.utils")
    uu.read_img = uu.resize = None
    sys.modules["fetal_net.utils.utils"] = uu
    nm = types.ModuleType("fetal_net.normalize")
    nm.normalize_data_storage = nm.normalize_data_storage_each = None
    sys.modules["fetal_net.normalize"] = nm
    spec = importlib.util.spec_from_file_location("fetal_net.data", os.path.join(REF, "fetal_net/data.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def make_data_file():
    D = load_reference_data_module()
    out = os.path.join(HERE, "pytables_data_golden.h5")
    if os.path.exists(out):
        os.remove(out)
    rs = np.random.RandomState(7)
    shapes = [(6, 5, 4), (7, 5, 3), (6, 6, 5)]                     # ragged: every subject has its own extent, as in the real data set
    hdf5_file, data_storage, truth_storage, mask_storage = D.create_data_file(out, n_samples=len(shapes))
    arrays = {}
    for i, sh in enumerate(shapes):
        vol = rs.randn(*sh) * 100 + 300
        truth = (rs.rand(*sh) > 0.6)
        mask = rs.rand(*sh) * 5
        D.add_data_to_storage(data_storage, truth_storage, mask_storage, [vol, truth, mask], np.uint8)
        arrays["data_%d" % i], arrays["truth_%d" % i], arrays["mask_%d" % i] = vol.astype(float), truth.astype(np.uint8), mask.astype(float)
    ids = ["fetus_001", "fetus_02", "f3"]
    hdf5_file.create_array(hdf5_file.root, 'subject_ids', obj=ids)        # data.py:65-66
    hdf5_file.close()
    # read back with PyTables itself: the expected values are what the reference would see
    f = D.open_data_file(out)
    for i in range(len(shapes)):
        for k in ("data", "truth", "mask"):
            got = getattr(f.root, k)[i]
            assert got.dtype == arrays["%s_%d" % (k, i)].dtype and np.array_equal(got, arrays["%s_%d" % (k, i)])
    f.close()
    arrays["subject_ids"] = np.array([s.encode() for s in ids])
    np.savez_compressed(os.path.join(HERE, "pytables_data_golden.npz"), **arrays)
    print("pytables_data_golden.h5:", os.path.getsize(out), "bytes")


def make_blosc_frames():
    """c-blosc frames through PyTables' own binding of the library (tables.utilsextension has none: use the HDF5 filter instead):
    write small chunked CArrays with the blosc filter and cut the raw chunk out of the file with h5py's read_direct_chunk"""
    import h5py
    out = {}
    rs = np.random.RandomState(1)
    cases = {
        "zeros_f8": np.zeros(4096, np.float64),
        "ramp_i4": np.arange(6000, dtype=np.int32),
        "noise_u1": rs.randint(0, 256, 3000).astype(np.uint8),                 # incompressible: blosc stores it memcpy'd
        "text_u1": np.frombuffer((b"the quick brown fox jumps over the lazy dog. " * 200), dtype=np.uint8).copy(),
        "smooth_f4": np.cumsum(rs.randn(6000)).astype(np.float32),
        "sparse_i8": (rs.rand(6000) > 0.97).astype(np.int64) * rs.randint(0, 1 << 40, 6000),
        "steps_i2": np.repeat(rs.randint(-300, 300, 300), 37).astype(np.int16),
        "big_ramp_i4": (np.arange(200000) // 7).astype(np.int32),               # 0.8 MB: several blosc blocks, each split per byte plane
        "refs_16": np.zeros(65536 * 2, np.uint64),                             # what a VLArray chunk looks like: 16-byte records, mostly zero
    }
    cases["refs_16"][:12] = rs.randint(1, 1 << 30, 12)
    tmp = os.path.join(HERE, "_blosc_tmp.h5")
    for shuffle in (True, False):
        for lvl in (1, 5, 9):
            f = tables.open_file(tmp, mode="w")
            for name, a in cases.items():
                f.create_carray(f.root, name, obj=a, chunkshape=a.shape, filters=tables.Filters(complevel=lvl, complib="blosc", shuffle=shuffle))
            f.close()
            with h5py.File(tmp, "r") as h:
                for name, a in cases.items():
                    mask, raw = h[name].id.read_direct_chunk((0,))
                    if mask != 0:            # the filter is optional: HDF5 stored this chunk unfiltered (blosc could not shrink it)
                        assert bytes(raw) == a.tobytes()
                        continue
                    key = "%s_s%d_l%d" % (name, int(shuffle), lvl)
                    out["frame_" + key] = np.frombuffer(raw, dtype=np.uint8).copy()
    os.remove(tmp)
    for name, a in cases.items():
        out["plain_" + name] = np.frombuffer(a.tobytes(), dtype=np.uint8).copy()
    np.savez_compressed(os.path.join(HERE, "blosc_frames_golden.npz"), **out)
    flags = sorted(set(int(v[2]) for k, v in out.items() if k.startswith("frame_")))
    print("blosc_frames_golden.npz:", sum(k.startswith("frame_") for k in out), "frames; header flag bytes seen:", [hex(x) for x in flags])


if __name__ == "__main__":
    make_data_file()
    make_blosc_frames()
