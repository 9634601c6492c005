"""Write tests/golden/adversarial_small.net + adversarial_small_expect.npz: a byte-level Torch7 checkpoint
`torch.save(filename, {D = MODEL_D, G = MODEL_G, opt = OPT, epoch = EPOCH})` (adversarial.lua:319-329) as a GPU run of the reference
would have left it, assembled HERE from the published Torch7 serialisation rules with nothing but `struct.pack` -- this script
does NOT import face_generator_amd.torch7_file (the writer under test's sibling) nor any other product code.

No Torch7 exists in this environment and the reference ships no checkpoint, so this is not a file written by Torch7; it is an
independent second implementation of the WRITER side (File.lua `writeObject`, Tensor / Storage `write`), so that the product's
READER is no longer checked only against its own writer.  What it reproduces of a real era file (upstream torch7 / nn / cunn /
cudnn.torch, Dec 2015 - Feb 2016; knowledge of those sources, none under /root/reference):

  * File.lua writeObject: int32 type tag (0 nil, 1 number = float64, 2 string = int32 length + bytes, 3 table, 4 torch object,
    5 boolean = int32); tables AND torch objects are numbered in write order (one shared counter); a second reference to the same
    object writes only tag + index; a torch object without its own write method serialises its fields as a NEW plain table (own index);
    strings "V 1" + class name in front of every first-time torch object.
  * Tensor write: int32 nDimension, int64 sizes, int64 strides, int64 storageOffset + 1, then the storage as an object (nil for a tensor
    without storage -- what prepareNetworkForSave's `torch.Tensor():typeAs(x)` leaves, nn_utils.lua:246-254).  Storage write: int64
    size + raw elements.  `long` = 8 bytes (64-bit Linux), little endian.
  * train.lua:151-152 `getParameters()`: every weight / bias of a net is a view (offset, size, stride) into ONE flat storage, every
    gradWeight / gradBias into a second one -- the storages are written once, at their first use, later tensors reference the index.
  * train.lua:139-145 NN_UTILS.activateCuda (nn_utils.lua:328-363): each net is `nn.Sequential{nn.Copy, <net on the GPU>, nn.Copy}`
    and its tensors are torch.CudaTensor / torch.CudaStorage (float32 payload).
  * Module fields of the era: nn.Linear {weight, bias, gradWeight, gradBias}; nn.View {size = LongStorage, numElements};
    nn.PReLU {nOutputPlane = 0, weight, gradWeight}; nn.SpatialUpSamplingNearest {scale_factor, inputSize, outputSize};
    cudnn.SpatialConvolution / nn.SpatialConvolution {nInputPlane, nOutputPlane, kW, kH, dW, dH, padW, padH, weight 4-D, ...};
    nn.SpatialBatchNormalization {eps, momentum, affine, running_mean, running_var, weight, bias, ...}; nn.SpatialDropout /
    nn.Dropout {p, noise, (v2)}; nn.SpatialAveragePooling {kW, kH, dW, dH, padW, padH, ceil_mode, count_include_pad, divide};
    every module {output, gradInput, _type, train}.  Lua's pairs() order is unspecified: keys are written sorted, which is NOT
    the order the product's writer uses.

The nets are the reference's topologies at reduced width (the full G is 10 MB): G = models.lua:57-81 from a 4x4 map with
8 -> 16 -> 8 -> 3 channels (16x16 output), D = the models.lua:382-416 pattern on 16x16 inputs.  The expected forward outputs in
evaluate mode (sample.lua's use of a loaded checkpoint) come from the test oracle on the same arrays.
    python tests/golden/make_t7_fixture.py
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import torch7_nn as O  # noqa: E402   (expected outputs only; the bytes below come from numpy arrays)

NIL, NUMBER, STRING, TABLE, TORCH, BOOLEAN = 0, 1, 2, 3, 4, 5


class Obj:
    """a torch class instance: class name + fields (dict) -- or, for tensors / storages, a payload writer"""

    def __init__(self, cls, fields=None):
        self.cls, self.fields = cls, fields


class Storage:
    def __init__(self, cls, arr):
        self.cls, self.arr = cls, np.ascontiguousarray(arr)


class Tensor:
    def __init__(self, cls, storage, offset, size, stride=None):
        self.cls, self.storage, self.offset, self.size = cls, storage, offset, tuple(size)
        if stride is None:
            stride, acc = [], 1
            for s in reversed(self.size):
                stride.insert(0, acc)
                acc *= s
        self.stride = tuple(stride)


class Emit:
    def __init__(self):
        self.out = []
        self.index = {}
        self.n = 0

    def i32(self, v): self.out.append(struct.pack("<i", v))
    def i64(self, v): self.out.append(struct.pack("<q", v))
    def f64(self, v): self.out.append(struct.pack("<d", v))

    def string(self, s):
        b = s.encode()
        self.i32(len(b))
        self.out.append(b)

    def obj(self, o):
        if o is None:
            self.i32(NIL)
        elif isinstance(o, bool):
            self.i32(BOOLEAN); self.i32(1 if o else 0)
        elif isinstance(o, (int, float)):
            self.i32(NUMBER); self.f64(float(o))
        elif isinstance(o, str):
            self.i32(STRING); self.string(o)
        elif isinstance(o, dict):
            if id(o) in self.index:
                self.i32(TABLE); self.i32(self.index[id(o)]); return
            self.n += 1
            self.index[id(o)] = self.n
            self.i32(TABLE); self.i32(self.n); self.i32(len(o))
            for k in sorted(o, key=lambda k: (isinstance(k, str), k)):          # numbers first (the array part), then names, sorted
                self.obj(k); self.obj(o[k])
        elif isinstance(o, (Obj, Storage, Tensor)):
            if id(o) in self.index:
                self.i32(TORCH); self.i32(self.index[id(o)]); return
            self.n += 1
            self.index[id(o)] = self.n
            self.i32(TORCH); self.i32(self.n); self.string("V 1"); self.string(o.cls)
            if isinstance(o, Storage):
                self.i64(o.arr.size)
                self.out.append(o.arr.tobytes())
            elif isinstance(o, Tensor):
                self.i32(len(o.size))
                for s in o.size: self.i64(s)
                for s in o.stride: self.i64(s)
                self.i64(o.offset + 1)
                self.obj(o.storage)
            else:
                self.obj(dict(o.fields))            # a NEW table every time: its own index
        else:
            raise TypeError(type(o))

    def bytes(self):
        return b"".join(self.out)


def lua_list(items):
    return {float(i + 1): v for i, v in enumerate(items)}


def build(rng):
    """-> (checkpoint table, expectations dict)"""
    T, St = "torch.CudaTensor", "torch.CudaStorage"
    exp = {}

    def empty():
        return Tensor(T, None, 0, ())                                   # torch.Tensor():typeAs(x): no dimensions, no storage

    def base(train=True):
        return {"output": empty(), "gradInput": empty(), "_type": T, "train": train}

    class Flat:                                                        # getParameters(): one storage for the parameters, one for the gradients
        def __init__(self):
            self.parts = []

        def view(self, arr):
            off = sum(a.size for a in self.parts)
            self.parts.append(np.asarray(arr, np.float32).reshape(-1))
            return off

        def finish(self):
            self.p = Storage(St, np.concatenate(self.parts))
            self.g = Storage(St, np.zeros(self.p.arr.size, np.float32))

    def net_objects(spec, flat):
        """spec: list of (class, fields-with-arrays); parameter arrays become views into flat.p / flat.g (bound after finish())"""
        mods, pending = [], []
        for cls, fields, params in spec:
            f = base()
            f.update(fields)
            for name, arr in params:
                off = flat.view(arr)
                pending.append((f, name, off, arr.shape))
            mods.append(Obj(cls, f))
        flat.finish()
        for f, name, off, shape in pending:
            f[name] = Tensor(T, flat.p, off, shape)
            f["grad" + name[0].upper() + name[1:]] = Tensor(T, flat.g, off, shape)
        return mods

    def u(shape, s):
        return rng.uniform(-s, s, shape).astype(np.float32)

    def conv(cls, ni, no, k, extra=None):
        s = 4.0 / np.sqrt(ni * k * k)          # (4 x Torch's reset range: the fixture's activations must spread)
        f = {"nInputPlane": ni, "nOutputPlane": no, "kW": k, "kH": k, "dW": 1, "dH": 1, "padW": (k - 1) // 2, "padH": (k - 1) // 2}
        if cls.startswith("cudnn"):
            f.update({"groups": 1, "iSize": Obj("torch.LongStorage", None)})
        else:
            f.update({"finput": empty(), "fgradInput": empty()})
        f.update(extra or {})
        return (cls, f, [("weight", u((no, ni, k, k), s)), ("bias", u((no,), s))])

    def bn(nf):
        return ("nn.SpatialBatchNormalization",
                {"eps": 1e-5, "momentum": 0.1, "affine": True, "nDim": 4,
                 "running_mean": None, "running_var": None, "save_mean": empty(), "save_std": empty()},
                [("weight", rng.uniform(0.5, 1.5, (nf,)).astype(np.float32)), ("bias", u((nf,), 0.2))])

    def prelu():
        return ("nn.PReLU", {"nOutputPlane": 0, "gradWeightBuf": empty(), "gradWeightBuf2": empty()},
                [("weight", np.array([rng.uniform(0.1, 0.4)], np.float32))])

    def linear(ni, no):
        s = 4.0 / np.sqrt(ni)
        return ("nn.Linear", {"addBuffer": empty()}, [("weight", u((no, ni), s)), ("bias", u((no,), s))])

    def longs(*v):
        return Storage("torch.LongStorage", np.array(v, np.int64))

    nz, c0, s0 = 10, 8, 4
    g_spec = [linear(nz, c0 * s0 * s0),
              ("nn.View", {"size": longs(c0, s0, s0), "numElements": c0 * s0 * s0}, []),
              prelu(),
              ("nn.SpatialUpSamplingNearest", {"scale_factor": 2, "inputSize": longs(0, 0, 0, 0), "outputSize": longs(0, 0, 0, 0)}, []),
              conv("cudnn.SpatialConvolution", c0, 16, 5), bn(16), prelu(),
              ("nn.SpatialUpSamplingNearest", {"scale_factor": 2, "inputSize": longs(0, 0, 0, 0), "outputSize": longs(0, 0, 0, 0)}, []),
              conv("cudnn.SpatialConvolution", 16, 8, 5), bn(8), prelu(),
              conv("cudnn.SpatialConvolution", 8, 3, 3),
              ("nn.Sigmoid", {}, [])]
    for spec in g_spec:                                                # cudnn's iSize: a LongStorage of 4 (any content)
        if "iSize" in spec[1]:
            spec[1]["iSize"] = longs(0, 0, 0, 0)
    d_spec = [conv("nn.SpatialConvolution", 3, 8, 3), prelu(),
              ("nn.SpatialDropout", {"p": 0.2, "noise": empty()}, []),
              ("nn.SpatialAveragePooling", {"kW": 2, "kH": 2, "dW": 2, "dH": 2, "padW": 0, "padH": 0, "ceil_mode": False,
                                            "count_include_pad": True, "divide": True}, []),
              conv("nn.SpatialConvolution", 8, 16, 3), prelu(),
              ("nn.SpatialDropout", {"p": 0.2, "noise": empty()}, []),
              ("nn.SpatialAveragePooling", {"kW": 2, "kH": 2, "dW": 2, "dH": 2, "padW": 0, "padH": 0, "ceil_mode": False,
                                            "count_include_pad": True, "divide": True}, []),
              ("nn.View", {"size": longs(16 * 4 * 4), "numElements": 16 * 4 * 4}, []),
              linear(16 * 4 * 4, 32), prelu(),
              ("nn.Dropout", {"p": 0.5, "noise": empty(), "v2": True, "inplace": False}, []),
              linear(32, 1),
              ("nn.Sigmoid", {}, [])]

    nets = {}
    for name, spec in (("G", g_spec), ("D", d_spec)):
        flat = Flat()
        mods = net_objects(spec, flat)
        # BatchNorm buffers live outside the flat vectors: own storages
        k = 0
        for m in mods:
            if m.cls == "nn.SpatialBatchNormalization":
                nf = m.fields["weight"].size[0]
                rm, rv = u((nf,), 0.5), rng.uniform(0.5, 2.0, (nf,)).astype(np.float32)
                m.fields["running_mean"] = Tensor(T, Storage(St, rm), 0, (nf,))
                m.fields["running_var"] = Tensor(T, Storage(St, rv), 0, (nf,))
                exp["%s_bn%d_running_mean" % (name, k)], exp["%s_bn%d_running_var" % (name, k)] = rm, rv
                k += 1
        inner = Obj("nn.Sequential", dict(base(), modules=lua_list(mods)))
        wrapped = Obj("nn.Sequential", dict(base(), modules=lua_list([
            Obj("nn.Copy", dict(base(), intype="torch.FloatTensor", outtype=T, dontCast=False)),
            inner,
            Obj("nn.Copy", dict(base(), intype=T, outtype="torch.FloatTensor", dontCast=False))])))
        wrapped.fields["_type"] = "torch.FloatTensor"
        nets[name] = wrapped
        exp["%s_flat" % name] = flat.p.arr.copy()
        exp["%s_classes" % name] = np.array([m.cls for m in mods])

    opt = {"save": "logs", "saveFreq": 30, "network": "", "noplot": True, "N_epoch": 1000, "batchSize": 32, "learningRate": 0.001,
           "seed": 1, "threads": 8, "gpu": 0, "noiseDim": nz, "window": 3, "scale": 16, "grayscale": False,
           "D_L1": 0, "D_L2": 1e-4, "G_L1": 0, "G_L2": 0, "D_clamp": 1, "G_clamp": 5, "D_iterations": 1, "G_iterations": 1,
           "D_maxAcc": 1.01, "D_optmethod": "adam", "G_optmethod": "adam", "geometry": {1.0: 3.0, 2.0: 16.0, 3.0: 16.0}}
    table = {"D": nets["D"], "G": nets["G"], "opt": opt, "epoch": 90}
    return table, exp


def oracle_nets(exp):
    """the same nets as oracle modules (for the expected evaluate-mode outputs)"""
    def fill(net, flat, prefix):
        off, k = 0, 0
        for m in net.modules:
            for name in ("weight", "bias"):
                a = getattr(m, name, None)
                if a is not None:
                    a[...] = flat[off:off + a.size].reshape(a.shape)
                    off += a.size
            if isinstance(m, O.SpatialBatchNormalization):
                m.running_mean[...] = exp["%s_bn%d_running_mean" % (prefix, k)]
                m.running_var[...] = exp["%s_bn%d_running_var" % (prefix, k)]
                k += 1
        assert off == flat.size
        return net
    r = np.random.default_rng(0)
    G = O.Sequential(O.Linear(10, 128, r), O.View(8, 4, 4), O.PReLU(), O.SpatialUpSamplingNearest(2),
                     O.SpatialConvolution(8, 16, 5, 5, 1, 1, 2, 2, r), O.SpatialBatchNormalization(16), O.PReLU(),
                     O.SpatialUpSamplingNearest(2), O.SpatialConvolution(16, 8, 5, 5, 1, 1, 2, 2, r), O.SpatialBatchNormalization(8),
                     O.PReLU(), O.SpatialConvolution(8, 3, 3, 3, 1, 1, 1, 1, r), O.Sigmoid())
    D = O.Sequential(O.SpatialConvolution(3, 8, 3, 3, 1, 1, 1, 1, r), O.PReLU(), O.SpatialDropout(0.2), O.SpatialAveragePooling(2, 2, 2, 2),
                     O.SpatialConvolution(8, 16, 3, 3, 1, 1, 1, 1, r), O.PReLU(), O.SpatialDropout(0.2), O.SpatialAveragePooling(2, 2, 2, 2),
                     O.View(256), O.Linear(256, 32, r), O.PReLU(), O.Dropout(0.5), O.Linear(32, 1, r), O.Sigmoid())
    return fill(G, exp["G_flat"], "G"), fill(D, exp["D_flat"], "D")


def main():
    rng = np.random.default_rng(20160215)
    table, exp = build(rng)
    e = Emit()
    e.obj(table)
    data = e.bytes()
    open(os.path.join(HERE, "adversarial_small.net"), "wb").write(data)
    G, D = oracle_nets(exp)
    G.evaluate(); D.evaluate()
    noise = rng.uniform(-1, 1, (6, 10)).astype(np.float32)
    images = G.forward(noise)
    exp["noise"], exp["G_images"] = noise, images.astype(np.float32)
    exp["D_out"] = D.forward(images.astype(np.float32)).astype(np.float32)
    exp["objects_written"] = np.array(e.n)
    np.savez_compressed(os.path.join(HERE, "adversarial_small_expect.npz"), **exp)
    print("adversarial_small.net: %d bytes, %d numbered objects; expect.npz keys: %s" % (len(data), e.n, sorted(exp)))


if __name__ == "__main__":
    main()
