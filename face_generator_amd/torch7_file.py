"""Torch7 binary serialisation (`torch.save` / `torch.load` of the reference era) -- SURVEY 8(f) rank 2.

The reference checkpoints are `torch.save(filename, {D = ..., G = ..., opt = OPT, epoch = EPOCH})`
(adversarial.lua:319-329, adversarial_c2f.lua:206-217) and are read back by `torch.load` (train.lua:114-129,
sample.lua:251-258).  The format is upstream torch7 `File.lua` + `Tensor.lua` / `Storage` writers (not under
/root/reference; restated from the published format, binary mode, little endian, `long` = 8 bytes):

    object    := int32 type, payload
    type 0 nil | 1 number: float64 | 2 string: int32 n, n bytes | 5 boolean: int32 0/1
    type 3 table : int32 index; first time only: int32 npairs, npairs x (object key, object value)
    type 4 torch : int32 index; first time only: string "V 1", string className, class payload
    tensor payload   : int32 nDim, nDim x int64 size, nDim x int64 stride, int64 storageOffset (1-based), object storage
    storage payload  : int64 n, n raw elements
    any other class  : object (a table holding the instance's fields)
    functions (types 6-8) are not supported (the reference's checkpoints hold none).

`index` numbers every table / torch object in write order so that shared storages (e.g. `getParameters()` views) and
repeated references are stored once.  HOST-SIDE ONLY: no device code; numpy arrays stand for tensors.

PARITY UNPINNED: there is no Torch7 in this container and the reference ships no checkpoint, so the writer/reader pair is
pinned only by round trips and by hand-assembled byte vectors of the published format (tests/test_torch7_file.py).
"""
import struct
from collections import OrderedDict

import numpy as np

TYPE_NIL, TYPE_NUMBER, TYPE_STRING, TYPE_TABLE, TYPE_TORCH, TYPE_BOOLEAN = 0, 1, 2, 3, 4, 5

_TENSOR_DTYPES = {
    "torch.FloatTensor": np.float32, "torch.DoubleTensor": np.float64, "torch.LongTensor": np.int64,
    "torch.IntTensor": np.int32, "torch.ShortTensor": np.int16, "torch.CharTensor": np.int8,
    "torch.ByteTensor": np.uint8, "torch.CudaTensor": np.float32,
}
_STORAGE_DTYPES = {k.replace("Tensor", "Storage"): v for k, v in _TENSOR_DTYPES.items()}
_DTYPE_TO_TENSOR = {np.dtype(np.float32): "torch.FloatTensor", np.dtype(np.float64): "torch.DoubleTensor",
                    np.dtype(np.int64): "torch.LongTensor", np.dtype(np.int32): "torch.IntTensor",
                    np.dtype(np.int16): "torch.ShortTensor", np.dtype(np.int8): "torch.CharTensor",
                    np.dtype(np.uint8): "torch.ByteTensor"}


class T7Object:
    """A torch class instance that is not a tensor/storage (nn modules, optim.ConfusionMatrix ...): typename + fields."""

    def __init__(self, typename, fields=None):
        self.typename = typename
        self.fields = fields if fields is not None else OrderedDict()

    def __getitem__(self, k):
        return self.fields[k]

    def get(self, k, default=None):
        return self.fields.get(k, default)

    def __repr__(self):
        return "T7Object(%s, %d fields)" % (self.typename, len(self.fields))


class T7Error(ValueError):
    pass


class LongStorage(tuple):
    """A tuple that is written as a bare `torch.LongStorage` (nn.View.size, SpatialUpSamplingNearest.inputSize ...)."""


# ------------------------------------------------------------------------------------------------------------ reader
class _Reader:
    def __init__(self, data):
        self.b = memoryview(data)
        self.p = 0
        self.memo = {}

    def _take(self, n):
        if self.p + n > len(self.b):
            raise T7Error("truncated Torch7 file (need %d bytes at offset %d of %d)" % (n, self.p, len(self.b)))
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def int32(self):
        return struct.unpack("<i", self._take(4))[0]

    def int64(self):
        return struct.unpack("<q", self._take(8))[0]

    def string(self):
        n = self.int32()
        if n < 0:
            raise T7Error("negative string length")
        return bytes(self._take(n)).decode("latin-1")

    def obj(self):
        t = self.int32()
        if t == TYPE_NIL:
            return None
        if t == TYPE_NUMBER:
            return struct.unpack("<d", self._take(8))[0]
        if t == TYPE_STRING:
            return self.string()
        if t == TYPE_BOOLEAN:
            return self.int32() != 0
        if t == TYPE_TABLE:
            idx = self.int32()
            if idx in self.memo:
                return self.memo[idx]
            tab = OrderedDict()
            self.memo[idx] = tab
            for _ in range(self.int32()):
                k = self.obj()
                v = self.obj()
                if isinstance(k, float) and k.is_integer():
                    k = int(k)
                tab[k] = v
            return tab
        if t == TYPE_TORCH:
            idx = self.int32()
            if idx in self.memo:
                return self.memo[idx]
            version = self.string()
            if version.startswith("V "):
                cls = self.string()
            else:                       # pre-versioning files: the first string already is the class name
                cls = version
            if cls in _STORAGE_DTYPES:
                n = self.int64()
                dt = np.dtype(_STORAGE_DTYPES[cls])
                arr = np.frombuffer(self._take(n * dt.itemsize), dtype=dt.newbyteorder("<")).astype(dt, copy=True)
                self.memo[idx] = arr
                return arr
            if cls in _TENSOR_DTYPES:
                nd = self.int32()
                size = [self.int64() for _ in range(nd)]
                stride = [self.int64() for _ in range(nd)]
                off = self.int64() - 1
                storage = self.obj()
                if nd == 0 or storage is None:
                    ten = np.zeros((0,), dtype=_TENSOR_DTYPES[cls])
                else:
                    ten = np.lib.stride_tricks.as_strided(storage[off:], shape=size,
                                                          strides=[s * storage.itemsize for s in stride])
                self.memo[idx] = ten        # a VIEW of its storage: tensors sharing a storage stay aliased
                return ten
            o = T7Object(cls)
            self.memo[idx] = o
            fields = self.obj()
            if isinstance(fields, OrderedDict):
                o.fields = fields
            else:
                o.fields = OrderedDict(value=fields)
            return o
        raise T7Error("unsupported Torch7 type tag %d at offset %d (functions are not supported)" % (t, self.p - 4))


def loads(data):
    r = _Reader(data)
    out = r.obj()
    return out


def load(path):
    with open(path, "rb") as f:
        return loads(f.read())


# ------------------------------------------------------------------------------------------------------------ writer
class _Writer:
    def __init__(self):
        self.out = []
        self.memo = {}       # id(object) -> index
        self.keep = []       # keep referenced temporaries alive so id() stays unique
        self.next = 1

    def int32(self, v):
        self.out.append(struct.pack("<i", int(v)))

    def int64(self, v):
        self.out.append(struct.pack("<q", int(v)))

    def string(self, s):
        b = s.encode("latin-1")
        self.int32(len(b))
        self.out.append(b)

    def _index(self, o, kind="obj"):
        """-> True if `o` was written before (only its index is emitted).  A 1-D array can be both a tensor and the
        storage under it, hence the kind."""
        k = (kind, id(o))
        if k in self.memo:
            self.int32(self.memo[k])
            return True
        self.memo[k] = self.next
        self.keep.append(o)
        self.int32(self.next)
        self.next += 1
        return False

    def _storage(self, base, cls):
        self.int32(TYPE_TORCH)
        if self._index(base, "storage"):
            return
        self.string("V 1")
        self.string(cls.replace("Tensor", "Storage"))
        self.int64(base.size)
        self.out.append(np.ascontiguousarray(base).astype(base.dtype.newbyteorder("<"), copy=False).tobytes())

    def tensor(self, a):
        cls = _DTYPE_TO_TENSOR.get(a.dtype)
        if cls is None:
            raise T7Error("no Torch7 tensor type for dtype %s" % a.dtype)
        self.int32(TYPE_TORCH)
        if self._index(a):
            return
        self.string("V 1")
        self.string(cls)
        if a.size == 0:
            self.int32(0)
            self.int64(1)
            self.int32(TYPE_NIL)
            return
        # the storage is the flat base buffer when `a` is a (positive-stride) view of one, else a contiguous copy
        base = a
        while isinstance(base.base, np.ndarray):
            base = base.base
        ok = base.ndim == 1 and base.flags.c_contiguous and base.dtype == a.dtype and all(s > 0 for s in a.strides)
        if ok:
            off = (a.__array_interface__["data"][0] - base.__array_interface__["data"][0]) // a.itemsize
            strides = [s // a.itemsize for s in a.strides]
        else:
            base = np.ascontiguousarray(a).reshape(-1)
            self.keep.append(base)
            off = 0
            strides = [s // a.itemsize for s in np.ascontiguousarray(a).strides]
        self.int32(a.ndim)
        for s in a.shape:
            self.int64(s)
        for s in strides:
            self.int64(s)
        self.int64(off + 1)
        self._storage(base, cls)

    def obj(self, o):
        if o is None:
            self.int32(TYPE_NIL)
        elif isinstance(o, (bool, np.bool_)):
            self.int32(TYPE_BOOLEAN)
            self.int32(1 if o else 0)
        elif isinstance(o, (int, float, np.integer, np.floating)):
            self.int32(TYPE_NUMBER)
            self.out.append(struct.pack("<d", float(o)))
        elif isinstance(o, str):
            self.int32(TYPE_STRING)
            self.string(o)
        elif isinstance(o, np.ndarray):
            self.tensor(o)
        elif isinstance(o, T7Object):
            self.int32(TYPE_TORCH)
            if self._index(o):
                return
            self.string("V 1")
            self.string(o.typename)
            self.obj(o.fields)
        elif isinstance(o, dict):
            self.int32(TYPE_TABLE)
            if self._index(o):
                return
            self.int32(len(o))
            for k, v in o.items():
                self.obj(k)
                self.obj(v)
        elif isinstance(o, LongStorage):
            self.int32(TYPE_TORCH)
            if self._index(o):
                return
            self.string("V 1")
            self.string("torch.LongStorage")
            self.int64(len(o))
            for v in o:
                self.int64(v)
        elif isinstance(o, (list, tuple)):          # Lua array: keys 1..n
            self.obj(OrderedDict((i + 1, v) for i, v in enumerate(o)))
        else:
            raise T7Error("cannot serialise %r to Torch7" % type(o))


def dumps(obj):
    w = _Writer()
    w.obj(obj)
    return b"".join(w.out)


def save(path, obj):
    data = dumps(obj)
    with open(path, "wb") as f:
        f.write(data)


def lua_array(tab):
    """OrderedDict with keys 1..n -> list (Lua array part), else the table unchanged."""
    if isinstance(tab, dict) and all(isinstance(k, int) for k in tab) and sorted(tab) == list(range(1, len(tab) + 1)):
        return [tab[i] for i in range(1, len(tab) + 1)]
    return tab
