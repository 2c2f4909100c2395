"""Reader for torch7 binary serialisations (`.t7`), numpy only.

`--mode original` of the reference loads its ten modules from torch7 files (WCT.py:36-46,
`trained_models/original_wct_models/{vgg_normalised,feature_invertor}_conv{k}_1.t7`) through
`torch.utils.serialization.load_lua` (model_original.py:5,26), which left torch in 1.0.  Only what that path
needs is restated here: the object stream of `torch.save(...)` in binary mode --

    int32 tag: 0 nil | 1 number (f64) | 2 string (int32 n + n bytes) | 3 table | 4 torch object | 5 boolean (int32)
               | 6, 7, 8 functions
    table / torch object / function: int32 reference index first; an index seen before is a back-reference
    table:        int32 count, then count x (key object, value object)
    torch object: string "V <n>" + class-name string (a version-less stream has the class name in place of "V <n>")
                  then the payload -- tensors: int32 ndim, int64 size[ndim], int64 stride[ndim], int64 1-based storage
                  offset, storage object; storages: int64 count + raw little-endian elements; every other class
                  (nn.Sequential, nn.SpatialConvolution ...): one object, the table of its fields

-- and the walk the reference does on the result: `model.get(i).weight / .bias` for fixed indices i
(utils.py:64-67, model_original.py:27-28, 59, 92-95, ... 561-573).  Those indices are exactly the convolution
modules of the `nn.Sequential` in order (pad / conv / relu / pool entries in between), so `sequential_convs`
returns the convolutions in sequence order and the caller pairs them with the module's layers in forward order.
"""
import struct
from typing import Any, Dict, List, Tuple

import numpy as np

_TENSORS = {"torch.FloatTensor": "<f4", "torch.DoubleTensor": "<f8", "torch.LongTensor": "<i8", "torch.IntTensor": "<i4",
            "torch.ShortTensor": "<i2", "torch.CharTensor": "i1", "torch.ByteTensor": "u1", "torch.CudaTensor": "<f4"}
_STORAGES = {k.replace("Tensor", "Storage"): v for k, v in _TENSORS.items()}


class T7Object:
    """A torch class instance that is not a tensor: `.torch_typename` + the table of its fields as attributes/items."""

    def __init__(self, typename: str, fields: Any):
        self.torch_typename = typename
        self.fields = fields if isinstance(fields, dict) else {"value": fields}

    def __getattr__(self, name):
        try:
            return self.__dict__["fields"][name]
        except KeyError:
            raise AttributeError("%s has no field %r" % (self.__dict__.get("torch_typename"), name))

    def __repr__(self):
        return "<t7 %s: %s>" % (self.torch_typename, ", ".join(str(k) for k in self.fields))


class T7Error(ValueError):
    pass


class _Reader:
    def __init__(self, buf: bytes, long_size: int = 8):
        self.buf, self.pos, self.objects, self.long_fmt = buf, 0, {}, "<q" if long_size == 8 else "<i"

    def _take(self, n: int) -> bytes:
        if n < 0 or self.pos + n > len(self.buf):
            raise T7Error("truncated torch7 stream at byte %d (wanted %d more)" % (self.pos, n))
        b = self.buf[self.pos:self.pos + n]
        self.pos += n
        return b

    def int(self) -> int:
        return struct.unpack("<i", self._take(4))[0]

    def long(self) -> int:
        return struct.unpack(self.long_fmt, self._take(struct.calcsize(self.long_fmt)))[0]

    def string(self) -> str:
        return self._take(self.int()).decode("latin-1")

    def obj(self) -> Any:
        tag = self.int()
        if tag == 0:
            return None
        if tag == 1:
            x = struct.unpack("<d", self._take(8))[0]
            return int(x) if x == int(x) and abs(x) < 2 ** 53 else x
        if tag == 2:
            return self.string()
        if tag == 5:
            return self.int() == 1
        if tag not in (3, 4, 6, 7, 8):
            raise T7Error("unknown torch7 type tag %d at byte %d (not a binary-mode .t7 file?)" % (tag, self.pos - 4))
        index = self.int()
        if index in self.objects:
            return self.objects[index]
        if tag in (6, 7, 8):  # a dumped Lua function: bytecode + upvalues; nothing on this path calls one
            self._take(self.int())
            self.objects[index] = fn = T7Object("function", {})
            fn.fields["upvalues"] = self.obj()
            return fn
        if tag == 3:
            table: Dict[Any, Any] = {}
            self.objects[index] = table
            for _ in range(self.int()):
                k = self.obj()
                table[k] = self.obj()
            return table
        version = self.string()
        name = self.string() if version.startswith("V ") else version
        if name in _TENSORS:
            ndim = self.int()
            size = [self.long() for _ in range(ndim)]
            stride = [self.long() for _ in range(ndim)]
            offset = self.long() - 1
            storage = self.obj()
            if storage is None or ndim == 0:
                arr = np.zeros((0,), _TENSORS[name])
            else:
                need = offset + 1 + sum((n - 1) * s for n, s in zip(size, stride)) if all(n > 0 for n in size) else 0
                if offset < 0 or need > storage.size:
                    raise T7Error("%s of size %s / stride %s / offset %d does not fit its storage of %d" % (name, size, stride, offset, storage.size))
                item = storage.dtype.itemsize
                arr = np.lib.stride_tricks.as_strided(storage[offset:], shape=size, strides=[s * item for s in stride], writeable=False)
            self.objects[index] = arr
            return arr
        if name in _STORAGES:
            n = self.long()
            dt = np.dtype(_STORAGES[name])
            arr = np.frombuffer(self._take(n * dt.itemsize), dt, n)
            self.objects[index] = arr
            return arr
        obj = T7Object(name, {})
        self.objects[index] = obj  # registered before the payload: fields may refer back to the object
        fields = self.obj()
        obj.fields = fields if isinstance(fields, dict) else {"value": fields}
        return obj


def load(path: str) -> Any:
    """The object a binary-mode `.t7` file holds (what load_lua returned, as plain python / numpy)."""
    with open(path, "rb") as f:
        buf = f.read()
    try:
        return _Reader(buf).obj()
    except T7Error:
        return _Reader(buf, long_size=4).obj()  # files written by a 32-bit torch (load_lua's long_size=4)


def _modules(seq: Any) -> List[Any]:
    mods = getattr(seq, "modules", None) if isinstance(seq, T7Object) else None
    if not isinstance(mods, dict):
        raise T7Error("not an nn container: %r" % (seq,))
    return [mods[i] for i in range(1, len(mods) + 1)]  # Lua arrays: keys 1..n


def sequential_convs(model: Any) -> List[Tuple[int, np.ndarray, np.ndarray]]:
    """(index as `model.get(index)` counts, weight [O,I,kh,kw] f32, bias [O] f32) of every convolution of an nn.Sequential,
    in order.  SpatialConvolutionMM keeps its weight as [O, I*kh*kw]; it is reshaped with the module's kH/kW."""
    out = []
    for i, m in enumerate(_modules(model)):
        if not (isinstance(m, T7Object) and m.torch_typename.split(".")[-1].startswith("SpatialConvolution")):
            continue
        w, b = np.asarray(m.weight, np.float32), np.asarray(m.bias, np.float32)
        if w.ndim == 2:
            kh, kw = int(m.kH), int(m.kW)
            w = w.reshape(w.shape[0], -1, kh, kw)
        if w.ndim != 4 or b.shape != (w.shape[0],):
            raise T7Error("module %d (%s): weight %s / bias %s" % (i, m.torch_typename, w.shape, b.shape))
        out.append((i, np.ascontiguousarray(w), np.ascontiguousarray(b)))
    return out
