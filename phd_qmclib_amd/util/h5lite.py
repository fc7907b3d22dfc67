"""A small h5py-shaped facade over the HDF5 C library (ctypes).

The reference stores results with h5py (`qmc_exec/io.py:76-208`); h5py is not
installed on this image's system interpreter, but libhdf5 itself is on the
machine.  `open_file` returns a real `h5py.File` when h5py is importable and
otherwise this facade, which implements exactly the subset the result files
need, with h5py's on-disk conventions so that either side can read the
other's files:

* groups (nested paths), links: `require_group`, `create_group`, `get`,
  `in`, `del`, `keys()`;
* datasets of float / int / uint / bool / fixed compound records and scalars,
  contiguous layout: `create_dataset(name, data=...)`, `dset[()]`;
* attributes: int, float, bool, str and 1-D arrays of numbers
  (`attrs.update`, `attrs.items`, ...).  bool is h5py's enum
  {FALSE = 0, TRUE = 1} over int8; str is a variable-length UTF-8 string.

Library lookup: $QMC_HDF5_LIB, the loader's search path, /opt/conda/lib.
"""
import ctypes as C
import ctypes.util
import glob
import os
import weakref

import numpy as np

__all__ = ['open_file', 'HDF5Unavailable', 'File', 'Group', 'Dataset']


class HDF5Unavailable(ImportError):
    """Neither h5py nor an HDF5 shared library could be loaded."""


hid_t = C.c_int64
herr_t = C.c_int
hsize_t = C.c_uint64
_P = C.c_void_p
_S = C.c_char_p

_SIGS = {
    'H5open': (herr_t, []),
    'H5Eset_auto2': (herr_t, [hid_t, _P, _P]),
    'H5free_memory': (herr_t, [_P]),
    'H5Fcreate': (hid_t, [_S, C.c_uint, hid_t, hid_t]),
    'H5Fopen': (hid_t, [_S, C.c_uint, hid_t]),
    'H5Fclose': (herr_t, [hid_t]),
    'H5Fflush': (herr_t, [hid_t, C.c_int]),
    'H5Gcreate2': (hid_t, [hid_t, _S, hid_t, hid_t, hid_t]),
    'H5Gopen2': (hid_t, [hid_t, _S, hid_t]),
    'H5Gclose': (herr_t, [hid_t]),
    'H5Gget_info': (herr_t, [hid_t, _P]),
    'H5Lexists': (C.c_int, [hid_t, _S, hid_t]),
    'H5Ldelete': (herr_t, [hid_t, _S, hid_t]),
    'H5Lget_name_by_idx': (C.c_ssize_t, [hid_t, _S, C.c_int, C.c_int, hsize_t,
                                         _P, C.c_size_t, hid_t]),
    'H5Screate': (hid_t, [C.c_int]),
    'H5Screate_simple': (hid_t, [C.c_int, _P, _P]),
    'H5Sclose': (herr_t, [hid_t]),
    'H5Sget_simple_extent_ndims': (C.c_int, [hid_t]),
    'H5Sget_simple_extent_dims': (C.c_int, [hid_t, _P, _P]),
    'H5Dcreate2': (hid_t, [hid_t, _S, hid_t, hid_t, hid_t, hid_t, hid_t]),
    'H5Dopen2': (hid_t, [hid_t, _S, hid_t]),
    'H5Dclose': (herr_t, [hid_t]),
    'H5Dwrite': (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, _P]),
    'H5Dread': (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, _P]),
    'H5Dget_space': (hid_t, [hid_t]),
    'H5Dget_type': (hid_t, [hid_t]),
    'H5Dvlen_reclaim': (herr_t, [hid_t, hid_t, hid_t, _P]),
    'H5Acreate2': (hid_t, [hid_t, _S, hid_t, hid_t, hid_t, hid_t]),
    'H5Aopen': (hid_t, [hid_t, _S, hid_t]),
    'H5Aopen_by_idx': (hid_t, [hid_t, _S, C.c_int, C.c_int, hsize_t, hid_t,
                               hid_t]),
    'H5Aclose': (herr_t, [hid_t]),
    'H5Awrite': (herr_t, [hid_t, hid_t, _P]),
    'H5Aread': (herr_t, [hid_t, hid_t, _P]),
    'H5Aget_space': (hid_t, [hid_t]),
    'H5Aget_type': (hid_t, [hid_t]),
    'H5Aget_name': (C.c_ssize_t, [hid_t, C.c_size_t, _P]),
    'H5Aexists': (C.c_int, [hid_t, _S]),
    'H5Adelete': (herr_t, [hid_t, _S]),
    'H5Tcopy': (hid_t, [hid_t]),
    'H5Tclose': (herr_t, [hid_t]),
    'H5Tcreate': (hid_t, [C.c_int, C.c_size_t]),
    'H5Tinsert': (herr_t, [hid_t, _S, C.c_size_t, hid_t]),
    'H5Tset_size': (herr_t, [hid_t, C.c_size_t]),
    'H5Tset_cset': (herr_t, [hid_t, C.c_int]),
    'H5Tset_strpad': (herr_t, [hid_t, C.c_int]),
    'H5Tget_class': (C.c_int, [hid_t]),
    'H5Tget_size': (C.c_size_t, [hid_t]),
    'H5Tget_sign': (C.c_int, [hid_t]),
    'H5Tis_variable_str': (C.c_int, [hid_t]),
    'H5Tenum_create': (hid_t, [hid_t]),
    'H5Tenum_insert': (herr_t, [hid_t, _S, _P]),
    'H5Tget_super': (hid_t, [hid_t]),
    'H5Tget_nmembers': (C.c_int, [hid_t]),
    'H5Tget_member_name': (_P, [hid_t, C.c_uint]),
    'H5Tget_member_offset': (C.c_size_t, [hid_t, C.c_uint]),
    'H5Tget_member_type': (hid_t, [hid_t, C.c_uint]),
}

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND, H5T_ENUM = 0, 1, 3, 6, 8
H5T_VARIABLE = C.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5S_SCALAR = 0

_lib = None


def _candidates():
    env = os.environ.get('QMC_HDF5_LIB')
    if env:
        yield env
    found = ctypes.util.find_library('hdf5')
    if found:
        yield found
    for pat in ('/opt/conda/lib/libhdf5.so*', '/usr/lib/x86_64-linux-gnu/'
                'libhdf5*.so*', '/usr/lib/x86_64-linux-gnu/hdf5/serial/'
                'libhdf5.so*'):
        for path in sorted(glob.glob(pat)):
            base = os.path.basename(path)
            if base.startswith(('libhdf5.so', 'libhdf5_serial.so')):
                yield path


def _load():
    global _lib
    if _lib is not None:
        return _lib
    errors = []
    for cand in _candidates():
        try:
            lib = C.CDLL(cand)
            for name, (res, args) in _SIGS.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            if lib.H5open() < 0:
                raise OSError('H5open failed')
            lib.H5Eset_auto2(0, None, None)    # no error-stack printing
            _lib = lib
            return lib
        except (OSError, AttributeError) as exc:
            errors.append(f'{cand}: {exc}')
    raise HDF5Unavailable('no usable HDF5 library (set QMC_HDF5_LIB); tried: '
                          + '; '.join(errors or ['nothing found']))


def _tid(name):
    """Predefined datatype id (the H5T_* macros are globals set by H5open)."""
    return hid_t.in_dll(_load(), name + '_g').value


_NATIVE = {
    'f8': 'H5T_NATIVE_DOUBLE', 'f4': 'H5T_NATIVE_FLOAT',
    'i1': 'H5T_NATIVE_INT8', 'i2': 'H5T_NATIVE_INT16',
    'i4': 'H5T_NATIVE_INT32', 'i8': 'H5T_NATIVE_INT64',
    'u1': 'H5T_NATIVE_UINT8', 'u2': 'H5T_NATIVE_UINT16',
    'u4': 'H5T_NATIVE_UINT32', 'u8': 'H5T_NATIVE_UINT64',
}


class _Type:
    """An HDF5 datatype id built for a numpy dtype (closed on exit when it is
    not one of the library's predefined ids)."""

    def __init__(self, dtype):
        lib = _load()
        self.own = []
        self.id = self._make(lib, np.dtype(dtype))

    def _make(self, lib, dt):
        if dt.kind == 'b':
            tid = lib.H5Tenum_create(_tid('H5T_NATIVE_INT8'))
            for name, val in ((b'FALSE', 0), (b'TRUE', 1)):
                v = C.c_int8(val)
                lib.H5Tenum_insert(tid, name, C.byref(v))
            self.own.append(tid)
            return tid
        if dt.fields:
            tid = lib.H5Tcreate(H5T_COMPOUND, dt.itemsize)
            self.own.append(tid)
            for fname, (fdt, off) in dt.fields.items():
                lib.H5Tinsert(tid, fname.encode(), off, self._make(lib, fdt))
            return tid
        key = dt.kind + str(dt.itemsize)
        if key not in _NATIVE:
            raise TypeError(f'dtype {dt} has no HDF5 equivalent here')
        return _tid(_NATIVE[key])

    def close(self):
        lib = _load()
        for tid in reversed(self.own):
            lib.H5Tclose(tid)
        self.own = []


def _str_type():
    lib = _load()
    tid = lib.H5Tcopy(_tid('H5T_C_S1'))
    lib.H5Tset_size(tid, H5T_VARIABLE)
    lib.H5Tset_cset(tid, H5T_CSET_UTF8)
    return tid


def _numpy_dtype(tid):
    """numpy dtype for a file datatype (None = variable-length string)."""
    lib = _load()
    cls = lib.H5Tget_class(tid)
    size = lib.H5Tget_size(tid)
    if cls == H5T_FLOAT:
        return np.dtype('f%d' % size)
    if cls == H5T_INTEGER:
        return np.dtype(('i%d' if lib.H5Tget_sign(tid) else 'u%d') % size)
    if cls == H5T_ENUM:
        # h5py's boolean convention; any other enum is read as its base type
        sup = lib.H5Tget_super(tid)
        base = _numpy_dtype(sup)
        lib.H5Tclose(sup)
        n = lib.H5Tget_nmembers(tid)
        names = []
        for i in range(n):
            p = lib.H5Tget_member_name(tid, i)
            names.append(C.cast(p, C.c_char_p).value)
            lib.H5free_memory(p)
        if sorted(names) == [b'FALSE', b'TRUE'] and base.itemsize == 1:
            return np.dtype(bool)
        return base
    if cls == H5T_STRING:
        if lib.H5Tis_variable_str(tid) > 0:
            return None
        return np.dtype('S%d' % size)
    if cls == H5T_COMPOUND:
        names, formats, offsets = [], [], []
        for i in range(lib.H5Tget_nmembers(tid)):
            p = lib.H5Tget_member_name(tid, i)
            names.append(C.cast(p, C.c_char_p).value.decode())
            lib.H5free_memory(p)
            mt = lib.H5Tget_member_type(tid, i)
            formats.append(_numpy_dtype(mt))
            lib.H5Tclose(mt)
            offsets.append(lib.H5Tget_member_offset(tid, i))
        return np.dtype({'names': names, 'formats': formats,
                         'offsets': offsets, 'itemsize': size})
    raise TypeError(f'unsupported HDF5 datatype class {cls}')


def _space_for(shape):
    lib = _load()
    if shape == ():
        return lib.H5Screate(H5S_SCALAR)
    dims = (hsize_t * len(shape))(*shape)
    return lib.H5Screate_simple(len(shape), dims, None)


def _shape_of(space):
    lib = _load()
    nd = lib.H5Sget_simple_extent_ndims(space)
    if nd <= 0:
        return ()
    dims = (hsize_t * nd)()
    lib.H5Sget_simple_extent_dims(space, dims, None)
    return tuple(int(d) for d in dims)


def _contiguous(arr):
    # (np.ascontiguousarray would turn a 0-d array into a 1-d one)
    return arr if arr.ndim == 0 else np.ascontiguousarray(arr)


def _check(rc, what):
    if rc < 0:
        raise OSError(f'HDF5: {what} failed')
    return rc


def _read(read_fn, obj, tid, space):
    """Shared by datasets and attributes: -> numpy array / scalar / str."""
    lib = _load()
    shape = _shape_of(space)
    dt = _numpy_dtype(tid)
    if dt is None:                         # variable-length strings
        n = int(np.prod(shape)) if shape else 1
        buf = (C.c_char_p * n)()
        mem = _str_type()
        _check(read_fn(obj, mem, buf), 'read')
        vals = [(b or b'').decode('utf-8') for b in buf]
        lib.H5Dvlen_reclaim(mem, space, 0, buf)
        lib.H5Tclose(mem)
        if shape == ():
            return vals[0]
        return np.array(vals, dtype=object).reshape(shape)
    mem = _Type(dt)
    out = np.empty(shape, dtype=dt)
    _check(read_fn(obj, mem.id, out.ctypes.data_as(_P)), 'read')
    mem.close()
    if dt.kind == 'S':
        out = out.astype(object)
        if shape == ():
            return out[()].decode('utf-8')
    if shape == ():
        return out[()]
    return out


class AttributeManager:
    def __init__(self, obj):
        self._o = obj

    def __setitem__(self, name, value):
        lib = _load()
        oid = self._o._id
        bname = name.encode()
        if lib.H5Aexists(oid, bname) > 0:
            lib.H5Adelete(oid, bname)
        if isinstance(value, (str, bytes)):
            if isinstance(value, bytes):
                value = value.decode('utf-8')
            tid = _str_type()
            sp = _space_for(())
            aid = _check(lib.H5Acreate2(oid, bname, tid, sp, 0, 0),
                         f'create attribute {name}')
            buf = (C.c_char_p * 1)(value.encode('utf-8'))
            _check(lib.H5Awrite(aid, tid, buf), f'write attribute {name}')
            lib.H5Aclose(aid); lib.H5Sclose(sp); lib.H5Tclose(tid)
            return
        if value is None:
            raise TypeError(f"attribute {name!r}: None has no HDF5 equivalent")
        arr = np.asarray(value)
        if arr.dtype.kind in 'OU':
            raise TypeError(f'attribute {name!r}: unsupported type '
                            f'{type(value).__name__}')
        arr = _contiguous(arr)
        ty = _Type(arr.dtype)
        sp = _space_for(arr.shape)
        aid = _check(lib.H5Acreate2(oid, bname, ty.id, sp, 0, 0),
                     f'create attribute {name}')
        _check(lib.H5Awrite(aid, ty.id, arr.ctypes.data_as(_P)),
               f'write attribute {name}')
        lib.H5Aclose(aid); lib.H5Sclose(sp); ty.close()

    def __getitem__(self, name):
        lib = _load()
        aid = lib.H5Aopen(self._o._id, name.encode(), 0)
        if aid < 0:
            raise KeyError(name)
        try:
            return self._read_open(aid)
        finally:
            lib.H5Aclose(aid)

    @staticmethod
    def _read_open(aid):
        lib = _load()
        tid, sp = lib.H5Aget_type(aid), lib.H5Aget_space(aid)
        try:
            return _read(lambda a, m, b: lib.H5Aread(a, m, b), aid, tid, sp)
        finally:
            lib.H5Tclose(tid); lib.H5Sclose(sp)

    def __contains__(self, name):
        return _load().H5Aexists(self._o._id, name.encode()) > 0

    def __delitem__(self, name):
        if _load().H5Adelete(self._o._id, name.encode()) < 0:
            raise KeyError(name)

    def get(self, name, default=None):
        return self[name] if name in self else default

    def keys(self):
        return [k for k, _ in self.items()]

    def items(self):
        lib = _load()
        out, i = [], 0
        while True:
            aid = lib.H5Aopen_by_idx(self._o._id, b'.', 0, 0, i, 0, 0)
            if aid < 0:
                break
            n = lib.H5Aget_name(aid, 0, None)
            buf = C.create_string_buffer(n + 1)
            lib.H5Aget_name(aid, n + 1, buf)
            out.append((buf.value.decode(), self._read_open(aid)))
            lib.H5Aclose(aid)
            i += 1
        return out

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.items())

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v


class _Object:
    _id = -1

    @property
    def attrs(self):
        return AttributeManager(self)


class Dataset(_Object):
    def __init__(self, did, file=None):
        self._id = did
        if file is not None:
            file._track(self)
        lib = _load()
        tid, sp = lib.H5Dget_type(did), lib.H5Dget_space(did)
        self.shape = _shape_of(sp)
        self.dtype = _numpy_dtype(tid)
        lib.H5Tclose(tid); lib.H5Sclose(sp)

    def __getitem__(self, key):
        lib = _load()
        tid, sp = lib.H5Dget_type(self._id), lib.H5Dget_space(self._id)
        try:
            val = _read(lambda d, m, b: lib.H5Dread(d, m, 0, 0, 0, b),
                        self._id, tid, sp)
        finally:
            lib.H5Tclose(tid); lib.H5Sclose(sp)
        if key == () or key is Ellipsis:
            return val
        return val[key]

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self[()])
        return a if dtype is None else a.astype(dtype)

    def __del__(self):
        if self._id >= 0 and _lib is not None:
            _lib.H5Dclose(self._id)
            self._id = -1


class Group(_Object):
    def __init__(self, gid, own=True, file=None):
        self._id, self._own = gid, own
        self._file = file
        if file is not None:
            file._track(self)

    # -- links ---------------------------------------------------------
    def _exists(self, path):
        lib = _load()
        cur = ''
        for part in [p for p in path.split('/') if p]:
            cur = f'{cur}/{part}' if cur else part
            if lib.H5Lexists(self._id, cur.encode(), 0) <= 0:
                return False
        return True

    def __contains__(self, path):
        return self._exists(path)

    def get(self, path, default=None):
        if not self._exists(path):
            return default
        lib = _load()
        did = lib.H5Dopen2(self._id, path.encode(), 0)
        if did >= 0:
            return Dataset(did, self._file)
        gid = lib.H5Gopen2(self._id, path.encode(), 0)
        if gid >= 0:
            return Group(gid, file=self._file)
        return default

    def __getitem__(self, path):
        obj = self.get(path)
        if obj is None:
            raise KeyError(path)
        return obj

    def __delitem__(self, path):
        if _load().H5Ldelete(self._id, path.encode(), 0) < 0:
            raise KeyError(path)

    def keys(self):
        lib = _load()
        info = (C.c_uint64 * 4)()           # H5G_info_t: nlinks is word 1
        _check(lib.H5Gget_info(self._id, info), 'H5Gget_info')
        names = []
        for i in range(int(info[1])):
            n = lib.H5Lget_name_by_idx(self._id, b'.', 0, 0, i, None, 0, 0)
            buf = C.create_string_buffer(n + 1)
            lib.H5Lget_name_by_idx(self._id, b'.', 0, 0, i, buf, n + 1, 0)
            names.append(buf.value.decode())
        return names

    def __iter__(self):
        return iter(self.keys())

    def create_group(self, path):
        if self._exists(path):
            raise ValueError(f'Unable to create group (name already exists): '
                             f'{path}')
        return self.require_group(path)

    def require_group(self, path):
        lib = _load()
        cur = ''
        for part in [p for p in path.split('/') if p]:
            cur = f'{cur}/{part}' if cur else part
            if lib.H5Lexists(self._id, cur.encode(), 0) <= 0:
                gid = _check(lib.H5Gcreate2(self._id, cur.encode(), 0, 0, 0),
                             f'create group {cur}')
                lib.H5Gclose(gid)
        gid = lib.H5Gopen2(self._id, path.encode(), 0)
        if gid < 0:
            raise TypeError(f'Incompatible object (not a group): {path}')
        return Group(gid, file=self._file)

    def create_dataset(self, name, data=None, shape=None, dtype=None):
        lib = _load()
        if data is None:
            data = np.zeros(shape, dtype=dtype or 'f8')
        if isinstance(data, str):
            tid = _str_type()
            sp = _space_for(())
            did = _check(lib.H5Dcreate2(self._id, name.encode(), tid, sp, 0,
                                        0, 0), f'create dataset {name}')
            buf = (C.c_char_p * 1)(data.encode('utf-8'))
            _check(lib.H5Dwrite(did, tid, 0, 0, 0, buf), 'write dataset')
            lib.H5Sclose(sp); lib.H5Tclose(tid)
            return Dataset(did, self._file)
        arr = _contiguous(np.asarray(data, dtype=dtype))
        if arr.dtype.kind in 'OU':
            raise TypeError(f'dataset {name!r}: object arrays are unsupported')
        if '/' in name.strip('/'):
            parent, _, leaf = name.strip('/').rpartition('/')
            return self.require_group(parent).create_dataset(leaf, data=arr)
        ty = _Type(arr.dtype)
        sp = _space_for(arr.shape)
        did = lib.H5Dcreate2(self._id, name.encode(), ty.id, sp, 0, 0, 0)
        if did < 0:
            lib.H5Sclose(sp); ty.close()
            raise ValueError(f'Unable to create dataset (name already exists '
                             f'or invalid): {name}')
        if arr.size:
            _check(lib.H5Dwrite(did, ty.id, 0, 0, 0, arr.ctypes.data_as(_P)),
                   f'write dataset {name}')
        lib.H5Sclose(sp); ty.close()
        return Dataset(did, self._file)

    def __del__(self):
        if self._own and self._id >= 0 and _lib is not None:
            _lib.H5Gclose(self._id)
            self._id = -1


class File(Group):
    """`h5py.File` stand-in: modes 'r', 'a' (default), 'w', 'r+'."""

    def __init__(self, path, mode='a'):
        lib = _load()
        bpath = os.fspath(path).encode()
        if mode == 'r':
            fid = lib.H5Fopen(bpath, H5F_ACC_RDONLY, 0)
        elif mode == 'r+':
            fid = lib.H5Fopen(bpath, H5F_ACC_RDWR, 0)
        elif mode == 'w':
            fid = lib.H5Fcreate(bpath, H5F_ACC_TRUNC, 0, 0)
        elif mode == 'a':
            if os.path.exists(path):
                fid = lib.H5Fopen(bpath, H5F_ACC_RDWR, 0)
            else:
                fid = lib.H5Fcreate(bpath, H5F_ACC_TRUNC, 0, 0)
        else:
            raise ValueError(f'invalid mode {mode!r}')
        if fid < 0:
            raise OSError(f"Unable to open file {os.fspath(path)!r} "
                          f"(mode {mode!r})")
        self._fid = fid
        self.filename = os.fspath(path)
        self.mode = mode
        self._children = []
        gid = _check(lib.H5Gopen2(fid, b'/', 0), 'open root group')
        super().__init__(gid)
        self._file = self

    def _track(self, obj):
        self._children.append(weakref.ref(obj))

    def flush(self):
        _load().H5Fflush(self._fid, 1)

    def close(self):
        """Closes the file and, like h5py, every object still open in it."""
        lib = _load()
        for ref in getattr(self, '_children', []):
            obj = ref()
            if obj is not None and obj._id >= 0:
                (lib.H5Dclose if isinstance(obj, Dataset)
                 else lib.H5Gclose)(obj._id)
                obj._id = -1
        self._children = []
        if self._id >= 0:
            lib.H5Gclose(self._id)
            self._id = -1
        if self._fid >= 0:
            lib.H5Fclose(self._fid)
            self._fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def open_file(path, mode='a'):
    """`h5py.File(path, mode)` when h5py is importable, else the ctypes
    facade over libhdf5 (raises HDF5Unavailable when neither exists)."""
    try:
        import h5py
        return h5py.File(path, mode)
    except ImportError:
        return File(path, mode)
