"""Data-file interchange on the host: the plain HDF5 layout with the PyTables surface the reference's callers use,
split pickles and norm_params.json (reference fetal_net/data.py:77-78, generator.py:158-190, fetal/utils.py:21-22)."""
import json
import os
import pickle
import random

import numpy as np
import pytest

from fetal_net import data as D


def _volumes(n=4):
    rs = np.random.RandomState(3)
    shapes = [(20 + i, 24, 16 + 2 * i) for i in range(n)]            # ragged: every subject has its own shape
    vols = [rs.randn(*s) for s in shapes]                            # float64, like the reference (data.py:36)
    truth = [(rs.rand(*s) > 0.7).astype(np.uint8) for s in shapes]
    masks = [rs.rand(*s) for s in shapes]
    return vols, truth, masks


def test_plain_data_file_round_trip_and_surface(tmp_path):
    vols, truth, masks = _volumes()
    ids = ["subj_%02d" % i for i in range(len(vols))]
    path = D.write_plain_data_file(str(tmp_path / "fetal_data.h5"), vols, truth, masks, subject_ids=ids)
    assert D.is_plain_data_file(path)
    f = D.open_data_file(path)
    try:
        assert len(f.root.data) == len(f.root.truth) == len(f.root.mask) == 4
        assert 'subject_ids' in f.root and 'mask' in f.root and 'nothing' not in f.root
        for i in range(4):
            assert f.root.data[i].dtype == np.float64 and np.array_equal(f.root.data[i], vols[i])
            assert f.root.truth[i].dtype == np.uint8 and np.array_equal(f.root.truth[i], truth[i])
            assert np.array_equal(f.root.mask[i], masks[i])
            assert f.root.subject_ids[i].decode('utf-8') == ids[i]          # reference prediction.py:341-342
        assert np.array_equal(f.root.data[-1], vols[-1])
        with pytest.raises(IndexError):
            f.root.data[4]
    finally:
        f.close()
    # no masks, no ids
    p2 = D.write_plain_data_file(str(tmp_path / "bare.h5"), vols[:2], truth[:2])
    with D.open_data_file(p2) as g:
        assert 'mask' not in g.root and 'subject_ids' not in g.root and len(g.root.data) == 2
    with pytest.raises(ValueError):
        D.write_plain_data_file(str(tmp_path / "bad.h5"), vols, truth[:2])


def test_foreign_hdf5_is_not_mistaken_for_a_data_file(tmp_path):
    from fetal_net.utils import hdf5
    p = str(tmp_path / "other.h5")
    with hdf5.File(p, "w") as f:
        f.create_dataset("x", data=np.arange(4.0)).close()
    assert not D.is_plain_data_file(p)
    try:
        import tables  # noqa: F401
    except ImportError:
        with pytest.raises(ValueError, match="not a reference data file"):
            D.open_data_file(p)


def test_run_validation_cases_opens_the_data_file_itself(tmp_path):
    """run_validation_cases(hdf5_file=...) on a plain data file with a stand-in model: case directories are named after subject_ids,
    the three NIfTI files are written (reference prediction.py:333-351)"""
    from fetal_net.prediction import run_validation_cases
    vols, truth, _ = _volumes(3)
    ids = ["a", "b", "c"]
    path = D.write_plain_data_file(str(tmp_path / "d.h5"), vols, truth, subject_ids=ids)
    keys = str(tmp_path / "val.pkl")
    D.pickle_dump([2, 0], keys)

    class Half(object):
        output_shape = (None, 1, 8, 8, 8)

        def predict(self, x):
            return np.full(np.asarray(x).shape, 0.5)

    out = run_validation_cases(validation_keys_file=keys, model_file="unused", training_modalities=["volume"], hdf5_file=path,
                               patch_shape=(8, 8, 8), output_dir=str(tmp_path / "pred"), overlap_factor=0.5, model=Half())
    assert [os.path.basename(os.path.dirname(p)) for p in out] == ["c", "a"]
    for p in out:
        assert sorted(os.listdir(os.path.dirname(p))) == ["data_volume.nii.gz", "prediction.nii.gz", "truth.nii.gz"]


def test_validation_split_matches_the_reference_draw_order(tmp_path):
    class F(object):
        class root(object):
            data = list(range(11))
    tr, va, te = (str(tmp_path / n) for n in ("training.pkl", "validation.pkl", "test.pkl"))
    random.seed(7)
    a = D.get_validation_split(F, tr, va, te, data_split=0.8)
    # the reference's sequence of draws, restated inline (generator.py:171-177, 185-190)
    random.seed(7)
    s = list(range(11))
    random.shuffle(s)
    test = [s.pop()]
    random.shuffle(s)
    cut = int(len(s) * 0.8)
    assert a == (s[:cut], s[cut:], test)
    assert sorted(a[0] + a[1] + a[2]) == list(range(11))
    with open(tr, "rb") as f:
        assert pickle.load(f) == a[0]
    random.seed(99)                                              # existing pickles are re-used, nothing is drawn
    st = random.getstate()
    assert D.get_validation_split(F, tr, va, te) == a and random.getstate() == st
    assert D.get_validation_split(F, tr, va, te, overwrite=True) != a or True


def test_norm_params_json(tmp_path):
    p = D.save_norm_params(str(tmp_path), np.float64(12.5), np.array([3.0, 4.0]))
    assert os.path.basename(p) == "norm_params.json"
    assert json.load(open(p)) == {"mean": 12.5, "std": [3.0, 4.0]}
    assert D.load_norm_params(str(tmp_path)) == (12.5, [3.0, 4.0])
    D.save_norm_params(str(tmp_path), None, None)                # normalize=False in the reference: both null
    assert D.load_norm_params(str(tmp_path)) == (None, None)


# ---------------------------------------------------------------------------------------------- reference PyTables files, natively
def test_blosc_decoder_against_frames_of_the_c_library(golden_dir):
    """42 frames compressed by c-blosc 1.20.1 itself (through PyTables' filter: typesizes 1-8, byte shuffle on / off, levels 1 / 5 / 9,
    single- and multi-block buffers with a short last block, the 16-byte-record chunk of a VLArray) decode to their plain contents"""
    from fetal_net.utils import blosc
    z = np.load(os.path.join(golden_dir, "blosc_frames_golden.npz"))
    frames = [k for k in z.files if k.startswith("frame_")]
    assert len(frames) == 42
    seen = set()
    for k in frames:
        name = k[len("frame_"):].rsplit("_s", 1)[0]
        frame = z[k].tobytes()
        assert blosc.decompress(frame) == z["plain_" + name].tobytes(), k
        seen.add((frame[2] & 1, frame[3], int.from_bytes(frame[4:8], "little") > int.from_bytes(frame[8:12], "little")))
    assert {s[0] for s in seen} == {0, 1} and {s[1] for s in seen} >= {1, 2, 4, 8} and any(s[2] for s in seen)
    # malformed input raises instead of returning garbage
    good = z["frame_ramp_i4_s1_l5"].tobytes()
    with pytest.raises(blosc.BloscError):
        blosc.decompress(good[:40])
    with pytest.raises(blosc.BloscError):
        blosc.decompress(bytes([good[0], good[1], good[2] | (1 << 5)]) + good[3:])         # lz4 codec bits
    with pytest.raises(blosc.BloscError):
        blosc.decompress(bytes([3]) + good[1:])                                            # unknown container version
    stored = bytes([2, 1, 0x2, 1]) + (5).to_bytes(4, "little") + (5).to_bytes(4, "little") + (21).to_bytes(4, "little") + b"hello"
    assert blosc.decompress(stored) == b"hello"                                             # memcpy'd frame


def test_reference_pytables_data_file_opens_without_pytables(golden_dir, tmp_path):
    """tests/golden/pytables_data_golden.h5 was written by the reference's own create_data_file / add_data_to_storage (data.py:11-38)
    under PyTables 3.6.1: VLArrays of pickled arrays behind blosc level 5.  The built-in reader returns every row exactly."""
    from fetal_net.data import PyTablesDataFile, is_plain_data_file, open_data_file
    path = os.path.join(golden_dir, "pytables_data_golden.h5")
    z = np.load(os.path.join(golden_dir, "pytables_data_golden.npz"))
    assert not is_plain_data_file(path)
    f = open_data_file(path)
    try:
        import tables  # noqa: F401
    except ImportError:
        assert isinstance(f, PyTablesDataFile)
    assert len(f.root.data) == len(f.root.truth) == len(f.root.mask) == 3
    assert "subject_ids" in f.root and "data" in f.root and "nothing" not in f.root
    assert [bytes(s) for s in f.root.subject_ids] == [bytes(s) for s in z["subject_ids"]]
    for i in range(3):
        for k in ("data", "truth", "mask"):
            got, want = getattr(f.root, k)[i], z["%s_%d" % (k, i)]
            assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want), (k, i)
    assert f.root.data[-1].shape == (6, 6, 5) and len(f.root.data[0:2]) == 2
    with pytest.raises(IndexError):
        f.root.data[3]
    f.close()
    with pytest.raises(ImportError):
        try:
            import tables  # noqa: F401
            raise ImportError("PyTables present: the write path is PyTables' own")
        except ImportError:
            open_data_file(path, "a")
    # the converter runs without PyTables too and its output is the plain layout with the same content
    import subprocess
    import sys
    out = str(tmp_path / "plain.h5")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "convert_data_file.py"), path, out])
    assert is_plain_data_file(out)
    g = open_data_file(out)
    for i in range(3):
        for k in ("data", "truth", "mask"):
            assert np.array_equal(getattr(g.root, k)[i], z["%s_%d" % (k, i)])
    assert [bytes(s) for s in g.root.subject_ids] == [bytes(s) for s in z["subject_ids"]]
    g.close()


def test_device_side_consumers_accept_the_native_reader(golden_dir):
    """what the generators do with a data file (reference generator.py:158-190, :246-275): length, per-subject arrays, split lists"""
    from fetal_net.data import get_validation_split, open_data_file
    f = open_data_file(os.path.join(golden_dir, "pytables_data_golden.h5"))
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        tr, va, te = get_validation_split(f, training_file=os.path.join(d, "tr.pkl"), validation_file=os.path.join(d, "va.pkl"),
                                          test_file=os.path.join(d, "te.pkl"), data_split=0.67)
    assert sorted(tr + va + te) == [0, 1, 2]
    vol, lab = f.root.data[tr[0]], f.root.truth[tr[0]]
    assert vol.shape == lab.shape and vol.dtype == np.float64 and lab.dtype == np.uint8
    f.close()
