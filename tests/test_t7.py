"""wct_hip/t7.py + model_zoo.load_t7_module: the torch7 checkpoints of --mode original (WCT.py:36-46).

The real files are not in the reference snapshot (README.md:26 points at a download) and load_lua left torch in 1.0, so
the reader is pinned on streams this file WRITES with the documented layout of torch7's binary serialiser (type tags,
reference indices, "V 1" + class name, tensors as size/stride/offset/storage) -- including what real files have and a
naive reader trips over: back-references to an object already read, tensors that view a shared storage at an offset,
SpatialConvolutionMM's 2-D weights, extra fields, nil / boolean / string values."""
import struct

import numpy as np
import pytest

from wct_hip import model_zoo, t7


class Writer:
    def __init__(self):
        self.b, self.next = bytearray(), 1

    def i32(self, v): self.b += struct.pack("<i", v)
    def i64(self, v): self.b += struct.pack("<q", v)
    def string_raw(self, s): self.i32(len(s)); self.b += s.encode()
    def nil(self): self.i32(0)
    def number(self, x): self.i32(1); self.b += struct.pack("<d", float(x))
    def string(self, s): self.i32(2); self.string_raw(s)
    def boolean(self, v): self.i32(5); self.i32(1 if v else 0)

    def new_index(self):
        self.next += 1
        return self.next - 1

    def table(self, items):
        """items: list of (key writer thunk, value writer thunk)"""
        self.i32(3); self.i32(self.new_index()); self.i32(len(items))
        for k, v in items:
            k(); v()

    def torch_header(self, cls, index=None):
        self.i32(4)
        index = self.new_index() if index is None else index
        self.i32(index); self.string_raw("V 1"); self.string_raw(cls)
        return index

    def backref(self, index): self.i32(4); self.i32(index)

    def storage(self, arr):
        idx = self.torch_header("torch.FloatStorage")
        self.i64(arr.size); self.b += arr.astype("<f4").tobytes()
        return idx

    def tensor(self, size, stride, offset0, storage_thunk):
        self.torch_header("torch.FloatTensor")
        self.i32(len(size))
        for v in size: self.i64(v)
        for v in stride: self.i64(v)
        self.i64(offset0 + 1)
        storage_thunk()

    def module(self, cls, fields):
        self.torch_header(cls)
        self.table(fields)


def write_module(path, kind, level, weights, key, mm=False):
    """An nn.Sequential laid out as the reference expects (pad / conv / relu triples, pool or unpool entries in between)."""
    w = Writer()
    layers = model_zoo.encoder_layers("original", level) if kind == "enc" else model_zoo.decoder_layers("original", level)
    mods = []

    def conv(name, k):
        wt, bs = weights["%s.%s.weight" % (key, name)], weights["%s.%s.bias" % (key, name)]
        def emit():
            # weight and bias view ONE storage: [junk(5) | weight | bias]; gradWeight is a back-reference to the weight tensor
            flat = np.concatenate([np.full(5, 7.0, np.float32), wt.reshape(-1), bs])
            st = {}
            def first_storage(): st["i"] = w.storage(flat)
            O, I = wt.shape[:2]
            wsize = [O, I * k * k] if mm else [O, I, k, k]
            wstride = [I * k * k, 1] if mm else [I * k * k, k * k, k, 1]
            tidx = {}
            def weight():
                tidx["w"] = w.next
                w.tensor(wsize, wstride, 5, first_storage)
            w.module("nn.SpatialConvolutionMM" if mm else "nn.SpatialConvolution", [
                (lambda: w.string("nInputPlane"), lambda: w.number(I)),
                (lambda: w.string("kH"), lambda: w.number(k)), (lambda: w.string("kW"), lambda: w.number(k)),
                (lambda: w.string("weight"), weight),
                (lambda: w.string("gradWeight"), lambda: w.backref(tidx["w"])),
                (lambda: w.string("bias"), lambda: w.tensor([O], [1], 5 + wt.size, lambda: w.backref(st["i"]))),
                (lambda: w.string("train"), lambda: w.boolean(False)),
                (lambda: w.string("_type"), lambda: w.string("torch.FloatTensor")),
                (lambda: w.string("finput"), lambda: w.nil()),
            ])
        return emit

    simple = lambda cls: (lambda: w.module(cls, [(lambda: w.string("train"), lambda: w.boolean(False))]))
    if kind == "enc":
        mods.append(conv("conv0", 1))
    for l in layers:
        mods += [simple("nn.SpatialReflectionPadding"), conv(l.name, 3), simple("nn.ReLU")]
        if l.pool_after:
            mods.append(simple("nn.SpatialMaxPooling"))
        if l.up_after:
            mods.append(simple("nn.SpatialUpSamplingNearest"))
    w.module("nn.Sequential", [
        (lambda: w.string("modules"), lambda: w.table([((lambda i=i: w.number(i + 1)), m) for i, m in enumerate(mods)])),
        (lambda: w.string("train"), lambda: w.boolean(False)),
    ])
    with open(path, "wb") as f:
        f.write(bytes(w.b))


@pytest.mark.parametrize("kind,level,mm", [("enc", 1, False), ("dec", 1, False), ("enc", 2, True), ("dec", 2, False), ("enc", 3, False), ("dec", 3, True)])
def test_t7_module_roundtrip(tmp_path, kind, level, mm):
    weights = model_zoo.synth_weights("original", 11)
    key = model_zoo.module_key(kind, level)
    path = str(tmp_path / ("%s.t7" % key))
    write_module(path, kind, level, weights, key, mm=mm)
    got = model_zoo.load_t7_module(path, kind, level)
    want = {k[len(key) + 1:]: v for k, v in weights.items() if k.startswith(key + ".")}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], want[k]), k
    seq = t7.load(path)
    assert seq.torch_typename == "nn.Sequential" and seq.train is False
    convs = t7.sequential_convs(seq)
    assert [c[0] for c in convs] == model_zoo.t7_indices(kind, level)
    first = seq.modules[convs[0][0] + 1]
    assert first.gradWeight is first.weight and first.finput is None and first._type == "torch.FloatTensor"


def test_t7_indices_are_the_references():
    # model_original.py:27-28, 59, 92-95, 135-137, 179-184, 232-236, 288-297, 360-368, 471-484, 561-573
    ref = {("enc", 1): [0, 2], ("dec", 1): [1], ("enc", 2): [0, 2, 5, 9], ("dec", 2): [1, 5, 8], ("enc", 3): [0, 2, 5, 9, 12, 16],
           ("dec", 3): [1, 5, 8, 12, 15], ("enc", 4): [0, 2, 5, 9, 12, 16, 19, 22, 25, 29], ("dec", 4): [1, 5, 8, 11, 14, 18, 21, 25, 28],
           ("enc", 5): [0, 2, 5, 9, 12, 16, 19, 22, 25, 29, 32, 35, 38, 42], ("dec", 5): [1, 5, 8, 11, 14, 18, 21, 24, 27, 31, 34, 38, 41]}
    for (kind, level), idx in ref.items():
        assert model_zoo.t7_indices(kind, level) == idx


def test_t7_errors(tmp_path):
    p = tmp_path / "bad.t7"
    p.write_bytes(b"\x09\x00\x00\x00rest")
    with pytest.raises(t7.T7Error):
        t7.load(str(p))
    weights = model_zoo.synth_weights("original", 11)
    path = str(tmp_path / "e2.t7")
    write_module(path, "enc", 2, weights, "e2")
    with pytest.raises(ValueError):          # an encoder-2 file offered as encoder 3: wrong module list
        model_zoo.load_t7_module(path, "enc", 3)
    data = open(path, "rb").read()
    (tmp_path / "cut.t7").write_bytes(data[: len(data) // 2])
    with pytest.raises(t7.T7Error):
        t7.load(str(tmp_path / "cut.t7"))
