"""Just enough HDF5 for the checkpoint interchange: groups, contiguous numeric datasets, string / numeric
attributes - the subset Keras 2.2 `model.save()` / `save_weights()` files use (reference fetal_net/training.py:31-32 writes them
through ModelCheckpoint, :45-86 reads them back).  h5py is not part of this stack; the HDF5 C library is (libhdf5 >= 1.10), so this
module binds it with ctypes.  The surface imitates the few h5py idioms Keras' saving code relies on:

    with File(path, "w") as f:
        f.attrs["backend"] = b"tensorflow"                      # bytes -> variable-length ASCII, str -> variable-length UTF-8
        f.attrs["layer_names"] = [b"conv3d_1", b"conv3d_2"]     # list of bytes -> fixed-length string array (numpy 'S' dtype)
        g = f.create_group("conv3d_1")
        g.create_dataset("conv3d_1/kernel:0", data=array)       # intermediate groups are created
    with File(path) as f:
        f.attrs["layer_names"]; f["conv3d_1"]["conv3d_1/kernel:0"][()]; list(f.keys())

Everything raises (OSError / KeyError / TypeError); nothing is silently skipped.  Chunking, references and compound types are not
written - Keras files use none of them.

For the reference's DATA files (PyTables VLArrays of pickled arrays behind the blosc filter, reference fetal_net/data.py:11-17) there
is a read path: `VLenBytes` reads the rows of a variable-length uint8 dataset, and `register_blosc_filter()` gives libhdf5 a decoder
for filter 32001 (a ctypes callback into fetal_net/utils/blosc.py) so that the library can expand the chunks that hold the rows'
heap references.
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
_lib = None


class _GInfo(C.Structure):
    _fields_ = [("storage_type", C.c_int), ("nlinks", hsize_t), ("max_corder", C.c_int64), ("mounted", C.c_uint), ("_pad", C.c_uint * 4)]


def _candidates():
    env = os.environ.get("FMRI_LIBHDF5")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for name in ("libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200", "libhdf5.so.310",
                 "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"):
        yield name


def available():
    try:
        lib()
        return True
    except OSError:
        return False


def lib():
    """the loaded library with argument / result types declared; OSError when there is no usable libhdf5"""
    global _lib
    if _lib is not None:
        return _lib
    errors = []
    h = None
    for name in _candidates():
        try:
            h = C.CDLL(name)
            break
        except OSError as e:
            errors.append("%s: %s" % (name, e))
    if h is None:
        raise OSError("no HDF5 C library found (set FMRI_LIBHDF5=/path/to/libhdf5.so); tried:\n  " + "\n  ".join(errors))
    maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
    h.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
    if (maj.value, mnr.value) < (1, 10):
        raise OSError("libhdf5 %d.%d.%d is too old (64-bit hid_t needs >= 1.10)" % (maj.value, mnr.value, rel.value))
    sig = {
        "H5open": (C.c_int, []), "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (C.c_int, [hid_t]), "H5Fflush": (C.c_int, [hid_t, C.c_int]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Gclose": (C.c_int, [hid_t]), "H5Gget_info": (C.c_int, [hid_t, C.POINTER(_GInfo)]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Lget_name_by_idx": (C.c_ssize_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (C.c_int, [hid_t]), "H5Iget_type": (C.c_int, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dclose": (C.c_int, [hid_t]), "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dvlen_reclaim": (C.c_int, [hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Screate": (hid_t, [C.c_int]), "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sclose": (C.c_int, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sget_simple_extent_type": (C.c_int, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (C.c_int, [hid_t]), "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tset_strpad": (C.c_int, [hid_t, C.c_int]), "H5Tset_cset": (C.c_int, [hid_t, C.c_int]),
        "H5Tget_native_type": (hid_t, [hid_t, C.c_int]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Aexists": (C.c_int, [hid_t, C.c_char_p]), "H5Aclose": (C.c_int, [hid_t]), "H5Adelete": (C.c_int, [hid_t, C.c_char_p]),
        "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]), "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]),
        "H5Aget_space": (hid_t, [hid_t]), "H5Aget_type": (hid_t, [hid_t]),
        "H5Aopen_by_idx": (hid_t, [hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, hid_t, hid_t]),
        "H5Aget_name": (C.c_ssize_t, [hid_t, C.c_size_t, C.c_char_p]),
        "H5Tvlen_create": (hid_t, [hid_t]), "H5Tget_super": (hid_t, [hid_t]),
        "H5Sselect_hyperslab": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Zregister": (C.c_int, [C.c_void_p]), "H5Zfilter_avail": (C.c_int, [C.c_int]),
        "H5allocate_memory": (C.c_void_p, [C.c_size_t, C.c_int]), "H5free_memory": (C.c_int, [C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(h, name)
        fn.restype, fn.argtypes = res, args
    if h.H5open() < 0:
        raise OSError("H5open failed")
    h.H5Eset_auto2(0, None, None)          # errors come back as negative ids / codes and are raised here, not printed by the library
    _lib = h
    return h


def _g(name):
    return hid_t.in_dll(lib(), name).value


def _check(v, what):
    if v < 0:
        raise OSError("HDF5: %s failed" % what)
    return v


_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING, _H5T_VLEN = 0, 1, 3, 9
_H5I_GROUP, _H5I_DATASET = 2, 5
_NATIVE = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g", "i1": "H5T_NATIVE_INT8_g", "u1": "H5T_NATIVE_UINT8_g",
           "i2": "H5T_NATIVE_INT16_g", "u2": "H5T_NATIVE_UINT16_g", "i4": "H5T_NATIVE_INT32_g", "u4": "H5T_NATIVE_UINT32_g",
           "i8": "H5T_NATIVE_INT64_g", "u8": "H5T_NATIVE_UINT64_g"}


def _c_order(a):
    """C-contiguous without np.ascontiguousarray's promotion of 0-d arrays to 1-d (scalars stay H5S_SCALAR)"""
    a = np.asarray(a)
    return a if a.flags.c_contiguous else a.copy(order="C")


def _native_of(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.bool_:
        dtype = np.dtype("u1")
    key = dtype.kind + str(dtype.itemsize)
    if key not in _NATIVE:
        raise TypeError("no HDF5 mapping for dtype %s" % dtype)
    return _g(_NATIVE[key])


def _space_of(shape):
    L = lib()
    if len(shape) == 0:
        return _check(L.H5Screate(0), "H5Screate")                                     # H5S_SCALAR
    dims = (hsize_t * len(shape))(*[int(s) for s in shape])
    return _check(L.H5Screate_simple(len(shape), dims, None), "H5Screate_simple")


def _shape_of(space):
    L = lib()
    n = _check(L.H5Sget_simple_extent_ndims(space), "H5Sget_simple_extent_ndims")
    if n == 0:
        return ()
    dims = (hsize_t * n)()
    L.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(d) for d in dims)


def _read(read_fn, obj, ftype, space):
    """shared by datasets and attributes: `read_fn(memtype, buffer)`"""
    L = lib()
    shape = _shape_of(space)
    count = int(np.prod(shape)) if shape else 1
    cls = L.H5Tget_class(ftype)
    if cls == _H5T_STRING:
        mem = _check(L.H5Tcopy(ftype), "H5Tcopy")
        try:
            if L.H5Tis_variable_str(ftype) > 0:
                buf = (C.c_char_p * count)()
                _check(read_fn(mem, buf), "read (variable-length strings)")
                vals = [bytes(buf[i]) if buf[i] is not None else b"" for i in range(count)]
                L.H5Dvlen_reclaim(mem, space, 0, buf)
                out = np.array(vals, dtype="S") if vals else np.array([], dtype="S1")
            else:
                size = L.H5Tget_size(ftype)
                raw = C.create_string_buffer(count * size)
                _check(read_fn(mem, raw), "read (fixed-length strings)")
                out = np.frombuffer(raw.raw, dtype="S%d" % size).copy()
        finally:
            L.H5Tclose(mem)
        return out.reshape(shape) if shape else out[0]
    if cls not in (_H5T_INTEGER, _H5T_FLOAT):
        raise TypeError("HDF5 type class %d is not supported by this reader" % cls)
    mem = _check(L.H5Tget_native_type(ftype, 1), "H5Tget_native_type")
    try:
        size = L.H5Tget_size(mem)
        kind = "f" if cls == _H5T_FLOAT else ("i" if L.H5Tget_sign(mem) == 1 else "u")
        out = np.empty(shape, dtype=np.dtype("%s%d" % (kind, size)))
        _check(read_fn(mem, out.ctypes.data_as(C.c_void_p)), "read")
    finally:
        L.H5Tclose(mem)
    return out if shape else out[()]


class Attributes(object):
    def __init__(self, owner):
        self._o = owner

    def __contains__(self, name):
        return lib().H5Aexists(self._o.id, name.encode()) > 0

    def keys(self):
        L = lib()
        out = []
        i = -1
        while True:
            i += 1
            a = L.H5Aopen_by_idx(self._o.id, b".", 0, 0, i, 0, 0)      # by name order; a negative id ends the walk
            if a < 0:
                break
            n = L.H5Aget_name(a, 0, None)
            buf = C.create_string_buffer(n + 1)
            L.H5Aget_name(a, n + 1, buf)
            L.H5Aclose(a)
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, name):
        L = lib()
        if name not in self:
            raise KeyError("no attribute %r on %s" % (name, self._o.name))
        a = _check(L.H5Aopen(self._o.id, name.encode(), 0), "H5Aopen")
        ftype, space = L.H5Aget_type(a), L.H5Aget_space(a)
        try:
            return _read(lambda mem, buf: L.H5Aread(a, mem, buf), a, ftype, space)
        finally:
            L.H5Tclose(ftype)
            L.H5Sclose(space)
            L.H5Aclose(a)

    def get(self, name, default=None):
        return self[name] if name in self else default

    def __setitem__(self, name, value):
        L = lib()
        if name in self:
            L.H5Adelete(self._o.id, name.encode())
        if isinstance(value, (bytes, str)):                           # scalar variable-length string, as h5py stores them
            raw = value if isinstance(value, bytes) else value.encode("utf8")
            t = _check(L.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
            L.H5Tset_size(t, C.c_size_t(-1).value)                     # H5T_VARIABLE
            if isinstance(value, str):
                L.H5Tset_cset(t, 1)                                    # H5T_CSET_UTF8
            space = _space_of(())
            a = _check(L.H5Acreate2(self._o.id, name.encode(), t, space, 0, 0), "H5Acreate2(%s)" % name)
            buf = (C.c_char_p * 1)(raw)
            rc = L.H5Awrite(a, t, buf)
            L.H5Aclose(a), L.H5Sclose(space), L.H5Tclose(t)
            _check(rc, "H5Awrite(%s)" % name)
            return
        arr = np.asarray(value)
        if arr.dtype.kind == "U":
            arr = np.char.encode(arr, "utf8")
        if arr.dtype.kind == "S":                                      # fixed-length, NUL-padded strings (numpy 'S')
            arr = _c_order(arr)
            t = _check(L.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
            L.H5Tset_size(t, max(arr.dtype.itemsize, 1))
            L.H5Tset_strpad(t, 1)                                      # H5T_STR_NULLPAD
            own_type = True
        else:
            arr = _c_order(arr.astype(np.uint8) if arr.dtype == np.bool_ else arr)
            t, own_type = _native_of(arr.dtype), False
        space = _space_of(arr.shape)
        a = _check(L.H5Acreate2(self._o.id, name.encode(), t, space, 0, 0), "H5Acreate2(%s)" % name)
        rc = L.H5Awrite(a, t, arr.ctypes.data_as(C.c_void_p)) if arr.size else 0
        L.H5Aclose(a), L.H5Sclose(space)
        if own_type:
            L.H5Tclose(t)
        _check(rc, "H5Awrite(%s)" % name)


class Dataset(object):
    def __init__(self, did, name):
        self.id, self.name = did, name
        self.attrs = Attributes(self)
        L = lib()
        space = L.H5Dget_space(did)
        self.shape = _shape_of(space)
        L.H5Sclose(space)

    def __getitem__(self, key):
        L = lib()
        ftype, space = L.H5Dget_type(self.id), L.H5Dget_space(self.id)
        try:
            out = _read(lambda mem, buf: L.H5Dread(self.id, mem, 0, 0, 0, buf), self, ftype, space)
        finally:
            L.H5Tclose(ftype)
            L.H5Sclose(space)
        return out if key == () or key is Ellipsis else out[key]

    def __array__(self, dtype=None, copy=None):
        a = self[()]
        return a if dtype is None else a.astype(dtype)

    def close(self):
        if self.id is not None:
            lib().H5Dclose(self.id)
            self.id = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------------------ blosc filter, VL rows
BLOSC_FILTER_ID = 32001
_H5Z_FLAG_REVERSE = 0x0100
_H5Z_FUNC = C.CFUNCTYPE(C.c_size_t, C.c_uint, C.c_size_t, C.POINTER(C.c_uint), C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_void_p))


class _H5ZClass2(C.Structure):
    _fields_ = [("version", C.c_int), ("id", C.c_int), ("encoder_present", C.c_uint), ("decoder_present", C.c_uint), ("name", C.c_char_p),
                ("can_apply", C.c_void_p), ("set_local", C.c_void_p), ("filter", _H5Z_FUNC)]


_filter_keepalive = []


def _blosc_filter(flags, cd_nelmts, cd_values, nbytes, buf_size, buf):
    """H5Z_func_t: decode only.  Returns the number of valid bytes in the new buffer, 0 on failure (HDF5's convention)."""
    try:
        if not (flags & _H5Z_FLAG_REVERSE):
            return 0                                        # writing through the blosc filter is not provided
        from . import blosc
        L = lib()
        plain = blosc.decompress(C.string_at(buf[0], nbytes))
        out = L.H5allocate_memory(max(len(plain), 1), 0)
        if not out:
            return 0
        C.memmove(out, plain, len(plain))
        L.H5free_memory(buf[0])
        buf[0] = out
        buf_size[0] = len(plain)
        return len(plain)
    except Exception:
        return 0


def register_blosc_filter():
    """make filter 32001 ('blosc', the one PyTables registers) known to this process's libhdf5 unless a plug-in already provides it"""
    L = lib()
    if _filter_keepalive or L.H5Zfilter_avail(BLOSC_FILTER_ID) > 0:
        return
    fn = _H5Z_FUNC(_blosc_filter)
    cls = _H5ZClass2(1, BLOSC_FILTER_ID, 0, 1, b"blosc", None, None, fn)
    _filter_keepalive.extend([fn, cls])
    _check(L.H5Zregister(C.byref(cls)), "H5Zregister(blosc)")


class _HVL(C.Structure):
    _fields_ = [("len", C.c_size_t), ("p", C.c_void_p)]


class VLenBytes(object):
    """rows of a 1-D dataset of H5T_VLEN { uint8 } - PyTables' VLArray with an ObjectAtom keeps one pickle per row this way"""

    def __init__(self, did, name):
        self.id, self.name = did, name
        self.attrs = Attributes(self)
        L = lib()
        space = L.H5Dget_space(did)
        self.shape = _shape_of(space)
        L.H5Sclose(space)
        ftype = L.H5Dget_type(did)
        base = L.H5Tget_super(ftype)
        ok = len(self.shape) == 1 and L.H5Tget_class(base) == _H5T_INTEGER and L.H5Tget_size(base) == 1
        L.H5Tclose(base)
        L.H5Tclose(ftype)
        if not ok:
            self.close()
            raise TypeError("%s: only 1-D variable-length byte datasets are supported" % name)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, i):
        i = int(i)
        if i < 0:
            i += self.shape[0]
        if not 0 <= i < self.shape[0]:
            raise IndexError(i)
        L = lib()
        register_blosc_filter()
        mem = _check(L.H5Tvlen_create(_g("H5T_NATIVE_UINT8_g")), "H5Tvlen_create")
        fspace = L.H5Dget_space(self.id)
        one = (hsize_t * 1)(1)
        mspace = _check(L.H5Screate_simple(1, one, None), "H5Screate_simple")
        try:
            _check(L.H5Sselect_hyperslab(fspace, 0, (hsize_t * 1)(i), None, one, None), "H5Sselect_hyperslab")
            row = _HVL()
            _check(L.H5Dread(self.id, mem, mspace, fspace, 0, C.byref(row)), "H5Dread(%s[%d]) - is the chunk filter available?" % (self.name, i))
            data = C.string_at(row.p, row.len) if row.len else b""
            L.H5Dvlen_reclaim(mem, mspace, 0, C.byref(row))
        finally:
            L.H5Sclose(mspace)
            L.H5Sclose(fspace)
            L.H5Tclose(mem)
        return data

    def close(self):
        if self.id is not None:
            lib().H5Dclose(self.id)
            self.id = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group(object):
    def __init__(self, gid, name, owns=True):
        self.id, self.name, self._owns = gid, name, owns
        self.attrs = Attributes(self)

    def __contains__(self, name):
        L = lib()
        cur = ""
        for part in [p for p in name.split("/") if p]:                 # H5Lexists wants every intermediate link to exist
            cur = part if not cur else cur + "/" + part
            if L.H5Lexists(self.id, cur.encode(), 0) <= 0:
                return False
        return True

    def keys(self):
        L = lib()
        info = _GInfo()
        _check(L.H5Gget_info(self.id, C.byref(info)), "H5Gget_info")
        names = []
        for i in range(int(info.nlinks)):
            n = _check(L.H5Lget_name_by_idx(self.id, b".", 0, 0, i, None, 0, 0), "H5Lget_name_by_idx")
            buf = C.create_string_buffer(n + 1)
            L.H5Lget_name_by_idx(self.id, b".", 0, 0, i, buf, n + 1, 0)
            names.append(buf.value.decode())
        return names

    def __iter__(self):
        return iter(self.keys())

    def __getitem__(self, name):
        L = lib()
        if name not in self:
            raise KeyError("no object %r in %s" % (name, self.name))
        oid = _check(L.H5Oopen(self.id, name.encode(), 0), "H5Oopen(%s)" % name)
        kind = L.H5Iget_type(oid)
        L.H5Oclose(oid)
        full = self.name.rstrip("/") + "/" + name
        if kind == _H5I_GROUP:
            return Group(_check(L.H5Gopen2(self.id, name.encode(), 0), "H5Gopen2"), full)
        if kind == _H5I_DATASET:
            did = _check(L.H5Dopen2(self.id, name.encode(), 0), "H5Dopen2")
            ftype = L.H5Dget_type(did)
            vlen = L.H5Tget_class(ftype) == _H5T_VLEN
            L.H5Tclose(ftype)
            return VLenBytes(did, full) if vlen else Dataset(did, full)
        raise TypeError("%s is neither a group nor a dataset" % full)

    def create_group(self, name):
        L = lib()
        parts = [p for p in name.split("/") if p]
        cur = ""
        gid = None
        for part in parts:
            cur = part if not cur else cur + "/" + part
            if gid is not None:
                L.H5Gclose(gid)
            if L.H5Lexists(self.id, cur.encode(), 0) > 0:
                gid = _check(L.H5Gopen2(self.id, cur.encode(), 0), "H5Gopen2(%s)" % cur)
            else:
                gid = _check(L.H5Gcreate2(self.id, cur.encode(), 0, 0, 0), "H5Gcreate2(%s)" % cur)
        return Group(gid, self.name.rstrip("/") + "/" + name)

    def require_group(self, name):
        return self.create_group(name)

    def create_dataset(self, name, shape=None, dtype=None, data=None):
        L = lib()
        if data is None:
            data = np.zeros(shape, dtype=dtype or np.float32)
        arr = _c_order(np.asarray(data) if dtype is None else np.asarray(data, dtype=dtype))
        if arr.dtype == np.bool_:
            arr = arr.astype(np.uint8)
        if "/" in name.strip("/"):
            parent = self.create_group(name.rsplit("/", 1)[0])
            parent.close()
        own_type = False
        if arr.dtype.kind == "U":
            arr = np.char.encode(arr, "utf8")
        if arr.dtype.kind == "S":                                      # fixed-length, NUL-padded byte strings (numpy 'S'; PyTables / h5py layout)
            arr = _c_order(arr)
            t = _check(L.H5Tcopy(_g("H5T_C_S1_g")), "H5Tcopy")
            L.H5Tset_size(t, max(arr.dtype.itemsize, 1))
            L.H5Tset_strpad(t, 1)                                      # H5T_STR_NULLPAD
            own_type = True
        else:
            t = _native_of(arr.dtype)
        space = _space_of(arr.shape)
        did = L.H5Dcreate2(self.id, name.encode(), t, space, 0, 0, 0)
        if did < 0:
            L.H5Sclose(space)
            if own_type:
                L.H5Tclose(t)
            raise OSError("HDF5: cannot create dataset %r in %s (name already in use?)" % (name, self.name))
        rc = L.H5Dwrite(did, t, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) if arr.size else 0
        L.H5Sclose(space)
        if own_type:
            L.H5Tclose(t)
        if rc < 0:
            L.H5Dclose(did)
            raise OSError("HDF5: H5Dwrite(%s) failed" % name)
        return Dataset(did, self.name.rstrip("/") + "/" + name)

    def close(self):
        if self.id is not None and self._owns:
            lib().H5Gclose(self.id)
        self.id = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class File(Group):
    """mode 'r' (default), 'r+' / 'a' (read-write, created when absent for 'a'), 'w' (truncate)"""

    def __init__(self, path, mode="r"):
        L = lib()
        p = os.fspath(path).encode()
        if mode == "w" or (mode == "a" and not os.path.exists(path)):
            fid = L.H5Fcreate(p, 2, 0, 0)                              # H5F_ACC_TRUNC
        elif mode in ("r+", "a"):
            fid = L.H5Fopen(p, 1, 0)                                   # H5F_ACC_RDWR
        elif mode == "r":
            fid = L.H5Fopen(p, 0, 0)
        else:
            raise ValueError("mode %r" % mode)
        if fid < 0:
            raise OSError("cannot open %s as an HDF5 file (mode %s)" % (path, mode))
        Group.__init__(self, fid, "/", owns=False)
        self.filename = os.fspath(path)

    def flush(self):
        lib().H5Fflush(self.id, 1)

    def close(self):
        if self.id is not None:
            lib().H5Fclose(self.id)
            self.id = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def is_hdf5(path):
    with open(path, "rb") as f:
        return f.read(8) == b"\x89HDF\r\n\x1a\n"
