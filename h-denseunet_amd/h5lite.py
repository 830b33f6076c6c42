"""Minimal read-only HDF5 reader for Keras weight files.

The public checkpoints of the reference (`densenet161_weights_tf.h5`, `model_best.hdf5`, README.md:25-33) are Keras
2.0.8 HDF5 files (K.engine/topology.py:2847-2873 / 3107-3167); h5py is not part of this image's Python, so this module
decodes the subset of the HDF5 file format those files use, in pure Python + numpy:

  superblock v0/v1 . old-style groups (v1 B-tree + symbol nodes + local heap) and compact link messages .
  object header v1 with continuation blocks . dataspace v1/v2 . datatypes: IEEE float, integers, fixed-length
  strings, variable-length strings (global heap) . contiguous and compact dataset layout (v3) . attribute messages
  v1/v2/v3

Anything else (chunked / compressed datasets, new-style superblocks, dense attribute storage) raises H5Error with the
feature named.  tests/test_h5_import.py reads files written by the real HDF5 library (tests/golden/make_keras_h5.py) --
fixed-length strings as h5py 2.x wrote them in the Keras 2.0.8 era and variable-length strings as h5py 3.x writes them.
"""
import struct
from collections import OrderedDict

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5Error(ValueError):
    pass


def _u(buf, off, n):
    return int.from_bytes(buf[off:off + n], "little")


class _Datatype:
    def __init__(self, buf, off):
        cv = buf[off]
        self.cls, self.version = cv & 0x0F, cv >> 4
        bits = buf[off + 1:off + 4]
        self.size = _u(buf, off + 4, 4)
        self.base = None
        self.np = None
        self.is_vlen_str = False
        if self.cls == 0:                                   # fixed point
            order = ">" if bits[0] & 1 else "<"
            self.np = np.dtype("%s%s%d" % (order, "i" if bits[0] & 8 else "u", self.size))
            self.nbytes = 8 + 4
        elif self.cls == 1:                                 # floating point (IEEE assumed for 2/4/8 bytes)
            order = ">" if bits[0] & 1 else "<"
            if self.size not in (2, 4, 8):
                raise H5Error("unsupported float size %d" % self.size)
            self.np = np.dtype("%sf%d" % (order, self.size))
            self.nbytes = 8 + 12
        elif self.cls == 3:                                 # fixed-length string
            self.np = np.dtype("S%d" % self.size)
            self.nbytes = 8
        elif self.cls == 9:                                 # variable length
            self.is_vlen_str = (bits[0] & 0x0F) == 1
            self.base = _Datatype(buf, off + 8)
            self.nbytes = 8 + self.base.nbytes
            if not self.is_vlen_str:
                raise H5Error("variable-length sequences are not supported (only variable-length strings)")
        else:
            raise H5Error("unsupported datatype class %d" % self.cls)


def _dataspace(buf, off, sl):
    ver, rank, flags = buf[off], buf[off + 1], buf[off + 2]
    if ver == 1:
        p = off + 8
    elif ver == 2:
        if buf[off + 3] == 2:
            return None                                     # null dataspace
        p = off + 4
    else:
        raise H5Error("unsupported dataspace version %d" % ver)
    return tuple(_u(buf, p + i * sl, sl) for i in range(rank))


class _Object:
    """one object header: messages decoded lazily into attrs / group links / dataset description"""

    def __init__(self, f, addr):
        self.f, self.addr = f, addr
        buf, so, sl = f.buf, f.so, f.sl
        if buf[addr:addr + 4] == b"OHDR":
            raise H5Error("version-2 object headers (libver='latest' files) are not supported")
        if buf[addr] != 1:
            raise H5Error("bad object header version %d at %d" % (buf[addr], addr))
        nmsg = _u(buf, addr + 2, 2)
        size = _u(buf, addr + 8, 4)
        blocks = [(addr + 16, size)]
        self.msgs = []
        while blocks and len(self.msgs) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(self.msgs) < nmsg:
                mtype, msize, mflags = _u(buf, p, 2), _u(buf, p + 2, 2), buf[p + 4]
                body = p + 8
                if mtype == 0x0010:                         # continuation
                    blocks.append((_u(buf, body, so), _u(buf, body + so, sl)))
                if mflags & 0x02:
                    raise H5Error("shared object header messages are not supported")
                self.msgs.append((mtype, body, msize))
                p = body + msize
        self._attrs = None
        self._links = None

    # ---- attributes
    @property
    def attrs(self):
        if self._attrs is None:
            self._attrs = OrderedDict()
            for mtype, p, size in self.msgs:
                if mtype == 0x000C:
                    name, val = self._attribute(p)
                    self._attrs[name] = val
                elif mtype == 0x0015:
                    buf = self.f.buf
                    if _u(buf, p + 2 + (2 if buf[p + 1] & 1 else 0), self.f.so) != self.f.undef:
                        raise H5Error("dense attribute storage is not supported")
        return self._attrs

    def _attribute(self, p):
        buf, sl = self.f.buf, self.f.sl
        ver = buf[p]
        nsz, tsz, ssz = _u(buf, p + 2, 2), _u(buf, p + 4, 2), _u(buf, p + 6, 2)
        q = p + 8
        if ver == 3:
            q += 1                                          # name character set
        elif ver not in (1, 2):
            raise H5Error("unsupported attribute message version %d" % ver)
        pad = (lambda n: (n + 7) & ~7) if ver == 1 else (lambda n: n)
        name = bytes(buf[q:q + nsz]).split(b"\0")[0].decode("utf8")
        q += pad(nsz)
        dt = _Datatype(buf, q)
        q += pad(tsz)
        shape = _dataspace(buf, q, sl)
        q += pad(ssz)
        return name, self.f._decode(dt, shape, q)

    # ---- group links
    @property
    def links(self):
        if self._links is None:
            f, buf, so = self.f, self.f.buf, self.f.so
            self._links = OrderedDict()
            for mtype, p, size in self.msgs:
                if mtype == 0x0011:                         # symbol table: B-tree + local heap
                    for name, addr in f._symbol_table(_u(buf, p, so), _u(buf, p + so, so)):
                        self._links[name] = addr
                elif mtype == 0x0006:                       # compact link message
                    flags = buf[p + 1]
                    q = p + 2
                    ltype = 0
                    if flags & 0x08:
                        ltype = buf[q]; q += 1
                    if flags & 0x04:
                        q += 8
                    if flags & 0x10:
                        q += 1
                    ln = 1 << (flags & 3)
                    nlen = _u(buf, q, ln); q += ln
                    name = bytes(buf[q:q + nlen]).decode("utf8"); q += nlen
                    if ltype != 0:
                        raise H5Error("soft / external links are not supported (%s)" % name)
                    self._links[name] = _u(buf, q, so)
                elif mtype == 0x0002:
                    raise H5Error("new-style groups with dense link storage are not supported")
        return self._links

    @property
    def is_dataset(self):
        return any(m[0] == 0x0008 for m in self.msgs)

    # ---- dataset
    def read(self):
        buf, so, sl = self.f.buf, self.f.so, self.f.sl
        dt = shape = layout = None
        for mtype, p, size in self.msgs:
            if mtype == 0x0003:
                dt = _Datatype(buf, p)
            elif mtype == 0x0001:
                shape = _dataspace(buf, p, sl)
            elif mtype == 0x0008:
                layout = p
            elif mtype == 0x000B:
                raise H5Error("filtered (compressed) datasets are not supported")
        if dt is None or layout is None:
            raise H5Error("not a dataset")
        ver, cls = buf[layout], buf[layout + 1]
        if ver != 3:
            raise H5Error("unsupported data layout message version %d" % ver)
        if cls == 0:
            data = layout + 4
        elif cls == 1:
            data = _u(buf, layout + 2, so)
            if data == self.f.undef:                        # never written: fill value (zeros)
                return np.zeros(shape or (), dt.np)
        else:
            raise H5Error("chunked datasets are not supported")
        return self.f._decode(dt, shape, data)


class Group:
    def __init__(self, f, obj, name):
        self._f, self._obj, self.name = f, obj, name

    @property
    def attrs(self):
        return self._obj.attrs

    def keys(self):
        return list(self._obj.links.keys())

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._obj.links:
                raise KeyError(path)
            obj = _Object(node._f, node._obj.links[part])
            full = node.name.rstrip("/") + "/" + part
            node = Dataset(node._f, obj, full) if obj.is_dataset else Group(node._f, obj, full)
        return node


class Dataset:
    def __init__(self, f, obj, name):
        self._f, self._obj, self.name = f, obj, name
        self._val = None

    @property
    def attrs(self):
        return self._obj.attrs

    def __array__(self, dtype=None, copy=None):
        if self._val is None:
            self._val = self._obj.read()
        return self._val if dtype is None else self._val.astype(dtype)

    @property
    def shape(self):
        return np.asarray(self).shape

    @property
    def dtype(self):
        return np.asarray(self).dtype

    def __getitem__(self, idx):
        return np.asarray(self)[idx]


class File(Group):
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = memoryview(fh.read())
        buf = self.buf
        if bytes(buf[:8]) != SIGNATURE:
            raise H5Error("%s: not an HDF5 file (user blocks are not supported)" % path)
        ver = buf[8]
        if ver not in (0, 1):
            raise H5Error("superblock version %d (libver='latest' files) is not supported" % ver)
        self.so, self.sl = buf[13], buf[14]
        self.undef = (1 << (8 * self.so)) - 1
        p = 24 + (4 if ver == 1 else 0)
        base = _u(buf, p, self.so)
        if base != 0:
            raise H5Error("non-zero base address is not supported")
        root_entry = p + 4 * self.so
        root_addr = _u(buf, root_entry + self.so, self.so)
        self._gheap = {}
        Group.__init__(self, self, _Object(self, root_addr), "/")

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    # ---- old-style group traversal
    def _symbol_table(self, btree, heap):
        buf, so, sl = self.buf, self.so, self.sl
        if bytes(buf[heap:heap + 4]) != b"HEAP":
            raise H5Error("bad local heap signature")
        hdata = _u(buf, heap + 8 + 2 * sl, so)
        out = []

        def walk(addr):
            if bytes(buf[addr:addr + 4]) != b"TREE":
                raise H5Error("bad B-tree signature")
            if buf[addr + 4] != 0:
                raise H5Error("unexpected B-tree node type")
            level, n = buf[addr + 5], _u(buf, addr + 6, 2)
            p = addr + 8 + 2 * so
            for i in range(n):
                child = _u(buf, p + sl + i * (sl + so), so)
                if level > 0:
                    walk(child)
                else:
                    if bytes(buf[child:child + 4]) != b"SNOD":
                        raise H5Error("bad symbol node signature")
                    nsym = _u(buf, child + 6, 2)
                    esz = 2 * so + 24
                    for k in range(nsym):
                        e = child + 8 + k * esz
                        noff, oaddr = _u(buf, e, so), _u(buf, e + so, so)
                        s = hdata + noff
                        t = s
                        while buf[t] != 0:
                            t += 1
                        out.append((bytes(buf[s:t]).decode("utf8"), oaddr))

        walk(btree)
        return out

    # ---- raw data -> numpy
    def _global_heap_object(self, addr, index):
        buf, sl = self.buf, self.sl
        if addr not in self._gheap:
            if bytes(buf[addr:addr + 4]) != b"GCOL":
                raise H5Error("bad global heap signature")
            size = _u(buf, addr + 8, sl)
            objs, p, end = {}, addr + 8 + sl, addr + size
            while p + 8 + sl <= end:
                idx, osz = _u(buf, p, 2), _u(buf, p + 8, sl)
                if idx == 0:
                    break
                objs[idx] = bytes(buf[p + 8 + sl:p + 8 + sl + osz])
                p += 8 + sl + ((osz + 7) & ~7)
            self._gheap[addr] = objs
        return self._gheap[addr][index]

    def _decode(self, dt, shape, data):
        buf, so = self.buf, self.so
        if shape is None:
            return None
        n = int(np.prod(shape)) if shape else 1
        if dt.cls == 9:
            esz = 4 + so + 4
            vals = []
            for i in range(n):
                e = data + i * esz
                ln, addr, idx = _u(buf, e, 4), _u(buf, e + 4, so), _u(buf, e + 4 + so, 4)
                vals.append(self._global_heap_object(addr, idx)[:ln] if ln else b"")
            arr = np.array(vals, dtype=object).reshape(shape)
            return arr if shape else vals[0]
        arr = np.frombuffer(buf, dtype=dt.np, count=n, offset=data).reshape(shape).copy()
        if dt.cls == 3 and not shape:
            return bytes(arr[()])
        return arr if shape or dt.cls == 3 else arr


def read_keras_weights(path):
    """{layer name: [arrays in the file's weight_names order]} of a Keras `save_weights` / `save` file
    (K.engine/topology.py:2847-2873; `Model.save` keeps the same group under /model_weights)."""
    f = File(path)
    g = f["model_weights"] if "layer_names" not in f.attrs and "model_weights" in f else f
    if "layer_names" not in g.attrs:
        raise H5Error("%s: no layer_names attribute (not a Keras weight file)" % path)

    def text(b):
        return (b if isinstance(b, bytes) else bytes(b)).rstrip(b"\0").decode("utf8")

    out = OrderedDict()
    for lname in [text(n) for n in np.asarray(g.attrs["layer_names"]).reshape(-1)]:
        lg = g[lname]
        names = [text(n) for n in np.asarray(lg.attrs["weight_names"]).reshape(-1)] if "weight_names" in lg.attrs else []
        out[lname] = [np.asarray(lg[w]) for w in names]
    return out


# =====================================================================================================================
# Writer: the same HDF5 subset, emitted natively (no h5py in this image), so that ModelCheckpoint / Model.save /
# save_weights hand the reference real Keras-layout files (K.engine/topology.py:2847-2873, K.models.py:31-170):
# superblock v0, old-style groups (local heap + symbol nodes + v1 B-tree of any depth), v1 object headers, contiguous
# little-endian datasets, v1 attribute messages with fixed-length (null-padded) strings -- what h5py 2.x wrote in the
# Keras 2.0.8 era.  tests/test_h5_export.py reads the files back with the reader above and, where the conda
# interpreter with the real HDF5 library exists, with h5py.
_UNDEF = 0xFFFFFFFFFFFFFFFF
_LEAF_K, _NODE_K = 4, 16            # HDF5 defaults: symbol nodes hold <= 2*4 entries, B-tree nodes <= 2*16 children


class WGroup:
    """in-memory group: .attrs (name -> bytes | str | list of str | numpy array), .children (name -> WGroup | ndarray)"""

    def __init__(self):
        self.attrs = OrderedDict()
        self.children = OrderedDict()

    def group(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            nxt = node.children.get(part)
            if nxt is None:
                nxt = node.children[part] = WGroup()
            if not isinstance(nxt, WGroup):
                raise H5Error("%s is a dataset" % part)
            node = nxt
        return node

    def dataset(self, path, arr):
        """create_dataset(name) with '/' in the name lands in nested groups, exactly as h5py resolves it"""
        parts = [p for p in path.split("/") if p]
        g = self.group("/".join(parts[:-1])) if len(parts) > 1 else self
        arr = np.asarray(arr)
        g.children[parts[-1]] = arr if arr.flags.c_contiguous else arr.copy(order="C")


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _dtype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (4, 8):
        exp_bits, man_bits = (8, 23) if dt.itemsize == 4 else (11, 52)
        bias = (1 << (exp_bits - 1)) - 1
        return (bytes([0x11, 0x20, dt.itemsize * 8 - 1, 0]) + struct.pack("<I", dt.itemsize) +
                struct.pack("<HHBBBBI", 0, dt.itemsize * 8, man_bits, exp_bits, 0, man_bits, bias))
    if dt.kind in "iu":
        return (bytes([0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0]) + struct.pack("<I", dt.itemsize) +
                struct.pack("<HH", 0, dt.itemsize * 8))
    if dt.kind == "S":
        return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", dt.itemsize)      # null-padded ASCII
    raise H5Error("cannot write dtype %s" % dt)


def _space_msg(shape):
    return bytes([1, len(shape), 0, 0, 0, 0, 0, 0]) + b"".join(struct.pack("<Q", int(d)) for d in shape)


def _attr_value(v):
    if isinstance(v, str):
        v = v.encode("utf8")
    if isinstance(v, bytes):
        return np.array(v, dtype="S%d" % max(len(v), 1))
    if isinstance(v, (list, tuple)):
        bs = [s.encode("utf8") if isinstance(s, str) else bytes(s) for s in v]
        return np.array(bs, dtype="S%d" % max([len(b) for b in bs] + [1]))
    v = np.asarray(v)
    return v if v.flags.c_contiguous else v.copy(order="C")


class _Emitter:
    def __init__(self):
        self.buf = bytearray(96)               # superblock placeholder

    def put(self, data):
        self.buf += b"\0" * (-len(self.buf) % 8)
        addr = len(self.buf)
        self.buf += data
        return addr

    def header(self, msgs):
        """v1 object header from [(type, body bytes)]"""
        body = b""
        for mtype, data in msgs:
            data = _pad8(data)
            if len(data) > 0xFFF8:
                raise H5Error("object header message of %d bytes exceeds the 64 KB limit of HDF5 (the same limit "
                              "h5py / Keras hit for very long attribute lists)" % len(data))
            body += struct.pack("<HHB3x", mtype, len(data), 0) + data
        return self.put(struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)

    def attr_msgs(self, attrs):
        out = []
        for name, v in attrs.items():
            a = _attr_value(v)
            nm = name.encode("utf8") + b"\0"
            dt, sp = _dtype_msg(a.dtype), _space_msg(a.shape)
            out.append((0x000C, struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) +
                        _pad8(sp) + a.tobytes()))
        return out

    def dataset(self, arr):
        arr = np.asarray(arr)
        if not arr.flags.c_contiguous:
            arr = arr.copy(order="C")
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        raw = arr.tobytes()
        data = self.put(raw) if raw else _UNDEF
        msgs = [(0x0001, _space_msg(arr.shape)), (0x0003, _dtype_msg(arr.dtype)),
                (0x0005, bytes([2, 2, 2, 0])),                                   # fill value v2: late alloc, undefined
                (0x0008, bytes([3, 1]) + struct.pack("<QQ", data, len(raw)))]
        return self.header(msgs)

    def group(self, g):
        """returns (object header address, B-tree address, local heap address)"""
        names = sorted(g.children, key=lambda s: s.encode("utf8"))
        addrs = {}
        for n in names:
            c = g.children[n]
            addrs[n] = self.group(c)[0] if isinstance(c, WGroup) else self.dataset(c)
        # local heap: offset 0 holds the empty string (the leftmost B-tree key)
        heap = bytearray(8)
        off = {}
        for n in names:
            off[n] = len(heap)
            heap += _pad8(n.encode("utf8") + b"\0")
        free_at = len(heap)
        heap += struct.pack("<QQ", 1, 32) + b"\0" * 16          # one free block (next = H5HL_FREE_NULL, size 32)
        hdata = self.put(bytes(heap))
        haddr = self.put(b"HEAP" + bytes(4) + struct.pack("<QQQ", len(heap), free_at, hdata))
        # symbol nodes
        snods = []                                               # (address, largest name offset)
        per = 2 * _LEAF_K
        for i in range(0, max(len(names), 1), per):
            chunk = names[i:i + per]
            ents = b"".join(struct.pack("<QQII16x", off[n], addrs[n], 0, 0) for n in chunk)
            ents += bytes(40 * (per - len(chunk)))
            a = self.put(b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk)) + ents)
            snods.append((a, off[chunk[-1]] if chunk else 0))
        # B-tree levels
        level, nodes = 0, snods
        while True:
            cap = 2 * _NODE_K
            groups = [nodes[i:i + cap] for i in range(0, len(nodes), cap)]
            size = 24 + (2 * cap + 1) * 8
            base = self.put(bytes(size * len(groups)))
            nxt = []
            left_key = 0
            for gi, grp in enumerate(groups):
                a = base + gi * size
                left = a - size if gi > 0 else _UNDEF
                right = a + size if gi + 1 < len(groups) else _UNDEF
                body = b"TREE" + struct.pack("<BBHQQ", 0, level, len(grp), left, right) + struct.pack("<Q", left_key)
                for child, key in grp:
                    body += struct.pack("<QQ", child, key)
                    left_key = key
                self.buf[a:a + len(body)] = body
                nxt.append((a, grp[-1][1]))
            if len(nxt) == 1:
                btree = nxt[0][0]
                break
            level, nodes = level + 1, nxt
        ohdr = self.header([(0x0011, struct.pack("<QQ", btree, haddr))] + self.attr_msgs(g.attrs))
        return ohdr, btree, haddr

    def finish(self, root):
        ohdr, btree, haddr = self.group(root)
        self.buf += b"\0" * (-len(self.buf) % 8)
        sb = (SIGNATURE + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack("<HHI", _LEAF_K, _NODE_K, 0) +
              struct.pack("<QQQQ", 0, _UNDEF, len(self.buf), _UNDEF) +
              struct.pack("<QQII", 0, ohdr, 1, 0) + struct.pack("<QQ", btree, haddr))
        assert len(sb) == 96
        self.buf[0:96] = sb
        return bytes(self.buf)


def write_file(path, root):
    """serialise the in-memory tree `root` (WGroup) to `path` atomically (temp file + rename)"""
    import os
    data = _Emitter().finish(root)
    tmp = "%s.tmp%d" % (path, os.getpid())
    with open(tmp, "wb") as fh:
        fh.write(data)
    os.replace(tmp, path)


def keras_weights_group(g, layers, keras_version="2.0.8", backend="tensorflow"):
    """K.engine/topology.py:2847-2873 save_weights_to_hdf5_group: `layers` = [(layer name, [(weight name, array)])]"""
    g.attrs["layer_names"] = [n for n, _ in layers]
    g.attrs["backend"] = backend
    g.attrs["keras_version"] = keras_version
    for lname, ws in layers:
        lg = g.group(lname)
        lg.attrs["weight_names"] = [wn for wn, _ in ws] if ws else np.zeros((0,), "S1")
        for wn, arr in ws:
            lg.dataset(wn, arr)
    return g


def read_nested_model_weights(path, model_group, swap="always"):
    """The author's multi-GPU loaders (K.engine/topology.py:3171-3247 `..._by_name_mulgpu`, :3250-3330
    `..._mulgpu_twomodelcombine`): the checkpoint was written from a `make_parallel` wrapper, so the real layers sit
    one level down, under the group of the wrapped model (`model_1`, `denseu161`, `auto3d_residual_conv`), and carry
    no weight_names attribute: the loaders list the HDF5 links (name order) and swap the first two, which turns
    (bias, kernel) / (beta, gamma, moving_mean, moving_variance) / (.._beta, .._gamma) into Keras' weight order.
    swap: 'always' (:3214, the mulgpu loader) or 'len2or4' (:3292-3293, the two-model loader).
    Returns {layer name: [arrays]}; raises KeyError when `model_group` is absent, as f[...] does there."""
    f = File(path)
    g = f["model_weights"] if "layer_names" not in f.attrs and "model_weights" in f else f
    mg = g[model_group]
    out = OrderedDict()
    for lname in mg.keys():
        lg = mg[lname]
        if not isinstance(lg, Group):
            continue
        names = lg.keys()
        if len(names) >= 2 and (swap == "always" or len(names) in (2, 4)):
            names[0], names[1] = names[1], names[0]
        out[lname] = [np.asarray(lg[w]) for w in names]
    return out
