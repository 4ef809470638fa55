"""
A small, dependency-free HDF5 writer/reader for the analysis output of the file handlers (core/output.py).

The reference writes its analysis sets with h5py (core/evaluator.py:366-700: groups `scales/` and `tasks/`, datasets
that grow along the first axis by one row per write, dimension scales attached to every task axis).  h5py and libhdf5
are not part of this image's python, so the same files are produced here directly from the published HDF5 file format
(specification version 1.1 structures, exactly the ones h5py's default `libver='earliest'` emits): version-0
superblock, version-1 object headers, symbol-table groups (v1 B-tree + local heap + symbol nodes), contiguous and
chunked (v1 B-tree indexed) dataset layouts, a global heap for variable-length strings and the object-reference lists
of the dimension-scale attributes (CLASS / NAME / REFERENCE_LIST on the scales, DIMENSION_LIST / DIMENSION_LABELS on
the tasks, as H5DS writes them).

Writing is two-phase: the whole tree (groups, datasets, attributes, scales) is declared first and serialized once by
`File.commit()`; after that only `Dataset.append(row)` (a new row of an extendible dataset: chunk allocation at the
end of the file, B-tree node, dataspace extent, end-of-file address updated in place) and `set_scalar_attr` (fixed-size
attribute values) touch the file, so it is a valid HDF5 file after every write.

`read(path)` parses the same subset back (and therefore also unfiltered files written by h5py with default settings):
used by `load_state` and by the 'append' mode of the handlers.

The encodings were cross-checked byte by byte against files written by h5py 3.3 / HDF5 1.10.6 and the files written
here are read back by that h5py in tests/test_output.py when such an interpreter is available.
"""

import os
import struct

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"
GROUP_LEAF_K = 4              # symbol nodes hold 2 K entries
GROUP_INTERNAL_K = 16
CHUNK_K = 32                  # chunk B-tree nodes hold 2 K entries (the superblock-v0 default)
GHEAP_MIN = 4096


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


# ---- datatypes ----------------------------------------------------------------------------------------------------
DT_F64 = bytes([0x11, 0x20, 0x3f, 0x00]) + struct.pack("<I", 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
DT_REF = bytes([0x17, 0, 0, 0]) + struct.pack("<I", 8)


def dt_int(size, signed=True):
    return bytes([0x10, 0x08 if signed else 0x00, 0, 0]) + struct.pack("<I", size) + struct.pack("<HH", 0, 8 * size)


DT_BOOL = bytes([0x18, 0x02, 0, 0]) + struct.pack("<I", 1) + dt_int(1) + _pad8(b"FALSE\0") + _pad8(b"TRUE\0") + b"\0\1"


def dt_fixed_str(n):
    return bytes([0x13, 0, 0, 0]) + struct.pack("<I", n)


def dt_vlen_str(utf8):
    return bytes([0x19, 0x01, 0x01 if utf8 else 0x00, 0]) + struct.pack("<I", 16) + dt_int(1, signed=False)


DT_VLEN_REF = bytes([0x19, 0, 0, 0]) + struct.pack("<I", 16) + DT_REF


def _member(name, offset, dt):
    return _pad8(name + b"\0") + struct.pack("<IB3xI4x4I", offset, 0, 0, 0, 0, 0, 0) + dt


DT_REFLIST = bytes([0x16, 0x02, 0, 0]) + struct.pack("<I", 16) + _member(b"dataset", 0, DT_REF) + \
    _member(b"dimension", 8, dt_int(4))


def np_datatype(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return DT_F64
    if dtype.kind in "iu":
        return dt_int(dtype.itemsize, dtype.kind == "i")
    if dtype == np.bool_:
        return DT_BOOL
    raise TypeError("h5lite: dtype %r is not supported" % (dtype,))


def space_scalar():
    return struct.pack("<BBB5x", 1, 0, 0)


def space_simple(dims, maxdims=None):
    maxdims = dims if maxdims is None else maxdims
    out = struct.pack("<BBB5x", 1, len(dims), 1)
    out += b"".join(struct.pack("<Q", d) for d in dims)
    out += b"".join(struct.pack("<Q", UNDEF if m is None else m) for m in maxdims)
    return out


def _message(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


# ==================================================================================================
# writer
# ==================================================================================================

class _Node:
    def __init__(self, file, name):
        self.file, self.name = file, name
        self.attrs = {}                     # name -> python value (encoded at commit)
        self.addr = 0
        self._attr_data_pos = {}            # name -> file offset of the attribute's data (after commit)

    # attribute encodings -----------------------------------------------------------------------------------------
    def _encode_attr(self, name, value, ctx):
        """-> (datatype, dataspace, data)"""
        if isinstance(value, tuple) and value and value[0] == "fixedstr":
            s = value[1].encode() + b"\0"
            return dt_fixed_str(len(s)), space_scalar(), s
        if isinstance(value, tuple) and value and value[0] == "reflist":
            data = b"".join(struct.pack("<Qi4x", ctx.addr(ds), dim) for (ds, dim) in value[1])
            return DT_REFLIST, space_simple((len(value[1]),)), data
        if isinstance(value, tuple) and value and value[0] == "vlen_refs":
            data = b""
            for refs in value[1]:
                if refs:
                    idx = ctx.heap_add(b"".join(struct.pack("<Q", ctx.addr(ds)) for ds in refs))
                    data += struct.pack("<IQI", len(refs), ctx.heap_addr, idx)
                else:
                    data += struct.pack("<IQI", 0, 0, 0)
            return DT_VLEN_REF, space_simple((len(value[1]),)), data
        if isinstance(value, tuple) and value and value[0] == "vlen_strs":
            data = b""
            for s in value[1]:
                if s:
                    idx = ctx.heap_add(s.encode())
                    data += struct.pack("<IQI", len(s.encode()), ctx.heap_addr, idx)
                else:
                    data += struct.pack("<IQI", 0, 0, 0)
            return dt_vlen_str(False), space_simple((len(value[1]),)), data
        if isinstance(value, str):
            raw = value.encode()
            idx = ctx.heap_add(raw)
            return dt_vlen_str(True), space_scalar(), struct.pack("<IQI", len(raw), ctx.heap_addr, idx)
        a = np.asarray(value)
        if a.dtype == np.bool_:
            pass
        elif a.dtype.kind in "iu":
            a = a.astype(np.int64)
        elif a.dtype.kind == "f":
            a = a.astype(np.float64)
        else:
            raise TypeError("h5lite: attribute %r of type %r is not supported" % (name, a.dtype))
        space = space_scalar() if a.ndim == 0 else space_simple(a.shape)
        return np_datatype(a.dtype), space, np.ascontiguousarray(a).tobytes()

    def _attr_messages(self, ctx, base):
        """attribute messages; records the file offset of every attribute's data (base = offset of the first message)"""
        out = b""
        for name, value in self.attrs.items():
            dt, sp, data = self._encode_attr(name, value, ctx)
            nm = name.encode() + b"\0"
            body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp)
            self._attr_data_pos[name] = base + len(out) + 8 + len(body)
            out += _message(0x000C, body + data)
        return out

    @staticmethod
    def _header(messages_bytes, nmsgs):
        return struct.pack("<BxHII4x", 1, nmsgs, 1, len(messages_bytes)) + messages_bytes


class Group(_Node):
    def __init__(self, file, name):
        super().__init__(file, name)
        self.children = {}

    def create_group(self, name):
        g = Group(self.file, name)
        self.children[name] = g
        return g

    def create_dataset(self, name, **kw):
        d = Dataset(self.file, name, **kw)
        self.children[name] = d
        return d

    def __getitem__(self, path):
        node = self
        for part in path.strip("/").split("/"):
            if part:
                node = node.children[part]
        return node

    def __contains__(self, name):
        return name in self.children

    # layout: header | B-tree node | heap header | heap data | symbol nodes
    def _sizes(self):
        names = sorted(self.children, key=lambda s: s.encode())
        heap = bytearray(8)                                   # offset 0: the empty string
        offs = []
        for n in names:
            offs.append(len(heap))
            heap += _pad8(n.encode() + b"\0")
        per = 2 * GROUP_LEAF_K
        nsnod = max(1, -(-len(names) // per))
        if nsnod > 2 * GROUP_INTERNAL_K:
            raise ValueError("h5lite: more than %d entries in group %r" % (per * 2 * GROUP_INTERNAL_K, self.name))
        return names, offs, bytes(heap), nsnod

    def serialize(self, ctx, addr):
        names, offs, heap, nsnod = self._sizes()
        nattr = len(self.attrs)
        hdr_len = 16 + 24
        attr_bytes = self._attr_messages(ctx, addr + hdr_len)
        btree_addr = addr + hdr_len + len(attr_bytes)
        btree_size = 24 + (2 * GROUP_INTERNAL_K + 1) * 8 + 2 * GROUP_INTERNAL_K * 8
        heap_addr = btree_addr + btree_size
        heap_data_addr = heap_addr + 32
        snod_addr = heap_data_addr + len(heap)
        snod_size = 8 + 2 * GROUP_LEAF_K * 40
        self.btree_addr, self.heap_addr = btree_addr, heap_addr
        msgs = _message(0x0011, struct.pack("<QQ", btree_addr, heap_addr)) + attr_bytes
        out = self._header(msgs, 1 + nattr)
        # B-tree node (type 0, level 0)
        per = 2 * GROUP_LEAF_K
        groups = [list(range(i, min(i + per, len(names)))) for i in range(0, max(len(names), 1), per)]
        bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, len(groups), UNDEF, UNDEF) + struct.pack("<Q", 0)
        for k, idxs in enumerate(groups):
            bt += struct.pack("<Q", snod_addr + k * snod_size)
            bt += struct.pack("<Q", offs[idxs[-1]] if idxs else 0)
        bt += b"\0" * (btree_size - len(bt))
        out += bt
        out += b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, heap_data_addr) + heap
        for idxs in groups:
            sn = b"SNOD" + struct.pack("<BxH", 1, len(idxs))
            for i in idxs:
                child = self.children[names[i]]
                if isinstance(child, Group):
                    sn += struct.pack("<QQI4xQQ", offs[i], ctx.addr(child), 1, child.btree_addr, child.heap_addr)
                else:
                    sn += struct.pack("<QQI4x16x", offs[i], ctx.addr(child), 0)
            sn += b"\0" * (snod_size - len(sn))
            out += sn
        return out


class Dataset(_Node):
    """Fixed datasets (data=...) are stored contiguously; datasets with maxshape[0] = None grow by rows (append)."""

    def __init__(self, file, name, shape=None, maxshape=None, dtype=np.float64, data=None):
        super().__init__(file, name)
        if data is not None:
            data = np.ascontiguousarray(data, dtype=dtype if shape is None and dtype is not None else None)
            data = data.astype(dtype) if dtype is not None else data
            shape = data.shape
        self.dtype = np.dtype(dtype)
        self.shape = tuple(int(s) for s in shape)
        self.data = data
        self.extendible = maxshape is not None and maxshape[0] is None
        if self.extendible:
            if self.shape[0] != 0:
                raise ValueError("h5lite: extendible datasets start empty")
            self.row_shape = self.shape[1:]
            self.row_bytes = int(np.prod(self.row_shape, dtype=np.int64)) * self.dtype.itemsize
            self.chunk_rows = 1 if self.row_bytes >= 4096 else max(1, 8192 // max(self.row_bytes, 1))
            if self.chunk_rows * self.row_bytes >= 2 ** 32:
                raise ValueError("h5lite: rows of 4 GiB or more are not supported")
            self.nrows = 0
            self.leaves = []               # [addr, [(first row, chunk addr), ...]]
            self.root_addr = None          # level-1 node once there are several leaves
        elif data is None:
            raise ValueError("h5lite: fixed datasets need their data at creation")
        self.dim_labels = {}
        self.dim_scales = {}
        self.scale_name = None
        self.scale_refs = []

    # ---- dimension scales (the H5DS conventions) ----------------------------------------------------------------------
    def make_scale(self, name):
        self.scale_name = name

    def set_label(self, axis, label):
        self.dim_labels[axis] = label

    def attach_scale(self, axis, scale):
        self.dim_scales.setdefault(axis, []).append(scale)
        scale.scale_refs.append((self, axis))

    @property
    def capacity(self):
        return (2 * CHUNK_K) ** 2 * self.chunk_rows if self.extendible else self.shape[0]

    def _finalize_attrs(self):
        if self.scale_name is not None:
            self.attrs["CLASS"] = ("fixedstr", "DIMENSION_SCALE")
            self.attrs["NAME"] = ("fixedstr", self.scale_name)
            if self.scale_refs:
                self.attrs["REFERENCE_LIST"] = ("reflist", list(self.scale_refs))
        if self.dim_labels:
            self.attrs["DIMENSION_LABELS"] = ("vlen_strs", [self.dim_labels.get(a, "") for a in range(len(self.shape))])
        if self.dim_scales:
            self.attrs["DIMENSION_LIST"] = ("vlen_refs", [self.dim_scales.get(a, []) for a in range(len(self.shape))])

    @property
    def _key_size(self):
        return 8 + 8 * (len(self.shape) + 1)

    @property
    def _node_size(self):
        return 24 + (2 * CHUNK_K + 1) * self._key_size + 2 * CHUNK_K * 8

    def serialize(self, ctx, addr):
        self._finalize_attrs()
        rank = len(self.shape)
        maxshape = ((None,) + self.shape[1:]) if self.extendible else self.shape
        space = _message(0x0001, space_simple(self.shape, maxshape))
        dtm = _message(0x0003, np_datatype(self.dtype), flags=1)
        if self.extendible:
            fill = _message(0x0005, bytes([2, 3, 0, 1]) + struct.pack("<I", 0), flags=1)
            lay_body_len = 3 + 8 + 4 * (rank + 1)
        else:
            fill = _message(0x0005, bytes([2, 2, 2, 1]) + struct.pack("<I", 0), flags=1)
            lay_body_len = 3 + 8 + 8
        lay_len = 8 + lay_body_len + (-lay_body_len % 8)
        pre = 16
        self._dims_pos = addr + pre + 8 + 8                            # first dimension of the dataspace message
        self._layout_addr_pos = addr + pre + len(space) + len(dtm) + len(fill) + 8 + 3
        attr_bytes = self._attr_messages(ctx, addr + pre + len(space) + len(dtm) + len(fill) + lay_len)
        hdr_total = pre + len(space) + len(dtm) + len(fill) + lay_len + len(attr_bytes)
        data_addr = addr + hdr_total
        if self.extendible:
            chunk_dims = (self.chunk_rows,) + self.row_shape + (self.dtype.itemsize,)
            layout = _message(0x0008, struct.pack("<BBBQ", 3, 2, rank + 1, data_addr) +
                              b"".join(struct.pack("<I", d) for d in chunk_dims))
            self.leaves = [[data_addr, []]]
            tail = self._node_bytes(0, [], UNDEF, UNDEF)
        else:
            raw = np.ascontiguousarray(self.data).tobytes()
            layout = _message(0x0008, struct.pack("<BBQQ", 3, 1, data_addr, len(raw)))
            tail = _pad8(raw)
        assert len(layout) == lay_len
        msgs = space + dtm + fill + layout + attr_bytes
        return self._header(msgs, 4 + len(self.attrs)) + tail

    # ---- chunk index ------------------------------------------------------------------------------------------------------
    def _key(self, nbytes, row):
        return struct.pack("<II", nbytes, 0) + struct.pack("<Q", row) + b"\0" * (8 * len(self.shape))

    def _final_key(self, row):
        return struct.pack("<II", 0, 0) + struct.pack("<Q", row) + \
            b"".join(struct.pack("<Q", d) for d in self.row_shape) + struct.pack("<Q", self.dtype.itemsize)

    def _node_bytes(self, level, entries, left, right):
        """entries: (first row, child address); chunk nodes of level 0 point at chunks, level 1 at leaves"""
        cb = self.chunk_rows * self.row_bytes
        out = b"TREE" + struct.pack("<BBHQQ", 1, level, len(entries), left, right)
        for (row, child) in entries:
            out += self._key(cb, row) + struct.pack("<Q", child)
        last = (entries[-1][0] + self.chunk_rows) if entries else 0
        if level > 0 and entries:
            last = self.leaves[-1][1][-1][0] + self.chunk_rows
        out += self._final_key(last)
        return out + b"\0" * (self._node_size - len(out))

    def append(self, row):
        f = self.file
        if not f.committed:
            raise RuntimeError("h5lite: commit() the file before appending rows")
        if self.nrows >= self.capacity:
            raise RuntimeError("h5lite: dataset %r is full (%d rows)" % (self.name, self.nrows))
        row = np.ascontiguousarray(row, dtype=self.dtype).reshape(self.row_shape)
        i = self.nrows
        if i % self.chunk_rows == 0:
            caddr = f._alloc(self.chunk_rows * self.row_bytes, zero=True)
            leaf = self.leaves[-1]
            if len(leaf[1]) >= 2 * CHUNK_K:
                new_addr = f._alloc(self._node_size)
                f._write(leaf[0], self._node_bytes(0, leaf[1], self.leaves[-2][0] if len(self.leaves) > 1 else UNDEF, new_addr))
                leaf = [new_addr, []]
                self.leaves.append(leaf)
                if self.root_addr is None:
                    self.root_addr = f._alloc(self._node_size)
                    f._write(self._layout_addr_pos, struct.pack("<Q", self.root_addr))
            leaf[1].append((i, caddr))
            left = self.leaves[-2][0] if len(self.leaves) > 1 else UNDEF
            f._write(leaf[0], self._node_bytes(0, leaf[1], left, UNDEF))
            if self.root_addr is not None:
                f._write(self.root_addr, self._node_bytes(1, [(lf[1][0][0], lf[0]) for lf in self.leaves], UNDEF, UNDEF))
        caddr = self.leaves[-1][1][-1][1]
        f._write(caddr + (i % self.chunk_rows) * self.row_bytes, row.tobytes())
        self.nrows = i + 1
        self.shape = (self.nrows,) + self.row_shape
        f._write(self._dims_pos, struct.pack("<Q", self.nrows))
        f._sync_eof()


class _Ctx:
    def __init__(self):
        self.addrs = {}
        self.heap_addr = 0
        self.heap_objs = []

    def addr(self, node):
        return self.addrs.get(id(node), 0)

    def heap_add(self, raw):
        self.heap_objs.append(raw)
        return len(self.heap_objs)


class File(Group):
    """HDF5 file under construction: declare groups / datasets / attributes, commit(), then append rows."""

    def __init__(self, path):
        Group.__init__(self, self, "/")
        self.path = str(path)
        self.committed = False
        self._fh = None
        self._eof = 0

    def _walk(self):
        out, stack = [], [self]
        while stack:
            n = stack.pop()
            out.append(n)
            if isinstance(n, Group):
                stack.extend(n.children[k] for k in sorted(n.children, reverse=True))
        return out

    def _layout(self, ctx, base):
        """serialize every object at consecutive addresses from `base`; children before parents so that the symbol
        table entries of sub-groups can cache their B-tree / heap addresses"""
        nodes = self._walk()
        blobs, pos = {}, base
        for n in reversed(nodes):
            ctx.addrs.setdefault(id(n), 0)
        order = list(reversed(nodes))
        for n in order:
            ctx.addrs[id(n)] = pos
            blob = _pad8(n.serialize(ctx, pos))
            blobs[id(n)] = blob
            pos += len(blob)
        return order, blobs, pos

    def commit(self):
        # pass 1: sizes (addresses and heap indices do not change any size)
        ctx = _Ctx()
        self._layout(ctx, 0)
        heap_need = 16 + sum(16 + len(_pad8(o)) for o in ctx.heap_objs) + 16
        heap_size = max(GHEAP_MIN, heap_need + (-heap_need % 8))
        sb_size = 96
        # pass 2: addresses of all objects (forward references need them), pass 3: final bytes
        ctx2 = _Ctx()
        ctx2.heap_addr = sb_size
        self._layout(ctx2, sb_size + heap_size)
        ctx3 = _Ctx()
        ctx3.heap_addr = sb_size
        ctx3.addrs = dict(ctx2.addrs)
        order, blobs, end = self._layout(ctx3, sb_size + heap_size)
        heap = b"GCOL" + struct.pack("<B3xQ", 1, heap_size)
        for k, o in enumerate(ctx3.heap_objs):
            heap += struct.pack("<HH4xQ", k + 1, 1, len(o)) + _pad8(o)
        free = heap_size - len(heap)
        heap += struct.pack("<HH4xQ", 0, 0, free) + b"\0" * (free - 16)
        root = ctx3.addr(self)
        sb = SIG + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, end, UNDEF)
        sb += struct.pack("<QQI4xQQ", 0, root, 1, self.btree_addr, self.heap_addr)
        assert len(sb) == sb_size
        tmp = self.path + ".tmp"
        with open(tmp, "wb") as fh:
            fh.write(sb + heap)
            for n in order:
                fh.write(blobs[id(n)])
        os.replace(tmp, self.path)
        self._fh = open(self.path, "r+b")
        self._eof = end
        self.committed = True

    # ---- in-place updates -----------------------------------------------------------------------------------------------
    def _alloc(self, n, zero=False):
        addr = self._eof
        self._eof += n + (-n % 8)
        if zero:
            self._fh.seek(addr)
            self._fh.write(b"\0" * (self._eof - addr))
        return addr

    def _write(self, pos, raw):
        self._fh.seek(pos)
        self._fh.write(raw)

    def _sync_eof(self):
        self._fh.seek(0, os.SEEK_END)
        size = self._fh.tell()
        if size < self._eof:
            self._fh.write(b"\0" * (self._eof - size))
        self._write(40, struct.pack("<Q", self._eof))

    def set_scalar_attr(self, node, name, value):
        """overwrite a fixed-size (int64 / float64 scalar) attribute of a committed file"""
        cur = node.attrs[name]
        raw = struct.pack("<q", int(value)) if isinstance(cur, (int, np.integer)) else struct.pack("<d", float(value))
        node.attrs[name] = type(cur)(value)
        self._write(node._attr_data_pos[name], raw)

    def flush(self):
        if self._fh is not None:
            self._fh.flush()

    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


# ==================================================================================================
# reader
# ==================================================================================================

class RDataset:
    def __init__(self, rd, name, msgs):
        self.rd, self.name = rd, name
        self.attrs = {}
        self.shape, self.dtype, self.layout = None, None, None
        for (t, body) in msgs:
            if t == 0x0001:
                self.shape = rd._space(body)
            elif t == 0x0003:
                self.dtype = rd._dtype(body)[0]
            elif t == 0x0008:
                self.layout = body
            elif t == 0x000C:
                k, v = rd._attr(body)
                self.attrs[k] = v

    def _chunks(self):
        ver, cls, nd = struct.unpack_from("<BBB", self.layout, 0)
        root = struct.unpack_from("<Q", self.layout, 3)[0]
        cdims = struct.unpack_from("<%dI" % nd, self.layout, 11)
        out = []
        ks = 8 + 8 * nd

        def walk(addr):
            b = self.rd.buf
            if addr == UNDEF or b[addr:addr + 4] != b"TREE":
                return
            ntype, level, n = struct.unpack_from("<BBH", b, addr + 4)
            p = addr + 24
            for _ in range(n):
                nbytes, mask = struct.unpack_from("<II", b, p)
                offs = struct.unpack_from("<%dQ" % nd, b, p + 8)
                child = struct.unpack_from("<Q", b, p + ks)[0]
                if mask:
                    raise NotImplementedError("h5lite: filtered chunks")
                if level == 0:
                    out.append((offs[:-1], child, nbytes))
                else:
                    walk(child)
                p += ks + 8
        walk(root)
        return cdims[:-1], out

    def read(self, index=None):
        """the whole array, or row `index` along the first axis"""
        ver, cls = struct.unpack_from("<BB", self.layout, 0)
        if ver != 3:
            raise NotImplementedError("h5lite: data layout version %d" % ver)
        n = int(np.prod(self.shape, dtype=np.int64))
        if cls == 1:
            addr, size = struct.unpack_from("<QQ", self.layout, 2)
            a = np.frombuffer(self.rd.buf, dtype=self.dtype, count=n, offset=addr).reshape(self.shape) if n else \
                np.zeros(self.shape, self.dtype)
            return np.array(a if index is None else a[index])
        if cls != 2:
            raise NotImplementedError("h5lite: data layout class %d" % cls)
        cdims, chunks = self._chunks()
        if index is not None:
            index = index % self.shape[0]
            lo, hi = index, index + 1
        else:
            lo, hi = 0, self.shape[0]
        out = np.zeros((hi - lo,) + tuple(self.shape[1:]), self.dtype)
        for offs, addr, nbytes in chunks:
            if offs[0] >= hi or offs[0] + cdims[0] <= lo:
                continue
            c = np.frombuffer(self.rd.buf, dtype=self.dtype, count=int(np.prod(cdims)), offset=addr).reshape(cdims)
            src = [slice(None)] * len(cdims)
            dst = [slice(None)] * len(cdims)
            for ax in range(len(cdims)):
                g0, g1 = offs[ax], min(offs[ax] + cdims[ax], self.shape[ax])
                if ax == 0:
                    g0, g1 = max(g0, lo), min(g1, hi)
                src[ax] = slice(g0 - offs[ax], g1 - offs[ax])
                dst[ax] = slice(g0 - (lo if ax == 0 else 0), g1 - (lo if ax == 0 else 0))
            out[tuple(dst)] = c[tuple(src)]
        return out if index is None else out[0]

    def __getitem__(self, key):
        return self.read()[key]


class RGroup:
    def __init__(self, rd, name, msgs):
        self.rd, self.name = rd, name
        self.attrs, self.links = {}, {}
        for (t, body) in msgs:
            if t == 0x0011:
                bt, heap = struct.unpack_from("<QQ", body, 0)
                self.links = rd._links(bt, heap)
            elif t == 0x000C:
                k, v = rd._attr(body)
                self.attrs[k] = v

    def keys(self):
        return list(self.links)

    def __contains__(self, k):
        return k in self.links

    def __getitem__(self, path):
        node = self
        for part in path.strip("/").split("/"):
            if part:
                node = node.rd._object(node.links[part], part)
        return node


class Reader(RGroup):
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        if b[:8] != SIG:
            raise ValueError("%s is not an HDF5 file" % path)
        ver = b[8]
        if ver > 1 or b[13] != 8 or b[14] != 8:
            raise NotImplementedError("h5lite: superblock version %d" % ver)
        p = 24 + (4 if ver == 1 else 0)
        root_entry = p + 32
        root_hdr = struct.unpack_from("<Q", b, root_entry + 8)[0]
        self._cache = {}
        RGroup.__init__(self, self, "/", self._messages(root_hdr))

    # ---- low-level parsing -----------------------------------------------------------------------------------------------
    def _messages(self, addr):
        b = self.buf
        ver, nmsgs, refc, size = struct.unpack_from("<BxHII", b, addr)
        if ver != 1:
            raise NotImplementedError("h5lite: object header version %d" % ver)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsgs:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsgs:
                t, sz, fl = struct.unpack_from("<HHB", b, p)
                body = b[p + 8:p + 8 + sz]
                if t == 0x0010:
                    blocks.append(struct.unpack_from("<QQ", body, 0))
                out.append((t, body))
                p += 8 + sz
        return out

    def _links(self, bt, heap):
        b = self.buf
        data_addr = struct.unpack_from("<Q", b, heap + 24)[0]
        links = {}

        def name_at(off):
            e = b.index(b"\0", data_addr + off)
            return b[data_addr + off:e].decode()

        def walk(addr):
            if b[addr:addr + 4] == b"TREE":
                ntype, level, n = struct.unpack_from("<BBH", b, addr + 4)
                for k in range(n):
                    walk(struct.unpack_from("<Q", b, addr + 24 + 8 + 16 * k)[0])
            elif b[addr:addr + 4] == b"SNOD":
                n = struct.unpack_from("<H", b, addr + 6)[0]
                for k in range(n):
                    off, hdr = struct.unpack_from("<QQ", b, addr + 8 + 40 * k)
                    links[name_at(off)] = hdr
        walk(bt)
        return links

    def _object(self, addr, name):
        if addr not in self._cache:
            msgs = self._messages(addr)
            is_group = any(t == 0x0011 for (t, _) in msgs)
            self._cache[addr] = RGroup(self, name, msgs) if is_group else RDataset(self, name, msgs)
        return self._cache[addr]

    def _space(self, body):
        ver, rank, flags = struct.unpack_from("<BBB", body, 0)
        off = 8 if ver == 1 else 4
        return tuple(struct.unpack_from("<%dQ" % rank, body, off)) if rank else ()

    def _dtype(self, body):
        """-> (numpy dtype or tag, encoded length)"""
        cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", body, 0)
        cls = cv & 0x0F
        if cls == 0:
            return np.dtype("%s%s%d" % (">" if b0 & 1 else "<", "i" if b0 & 8 else "u", size)), 12
        if cls == 1:
            return np.dtype("%sf%d" % (">" if b0 & 1 else "<", size)), 20
        if cls == 3:
            return ("str", size), 8
        if cls == 7:
            return np.dtype("<u8"), 8
        if cls == 8:
            base, n = self._dtype(body[8:])
            return (np.dtype(np.bool_) if size == 1 else base), None
        if cls == 9:
            return ("vlen_str" if (b0 & 0x0F) == 1 else "vlen", None), None
        return ("opaque", size), None

    def _gheap(self, addr, index):
        b = self.buf
        size = struct.unpack_from("<Q", b, addr + 8)[0]
        p = addr + 16
        while p < addr + size:
            idx, ref, n = struct.unpack_from("<HH4xQ", b, p)
            if idx == 0:
                break
            if idx == index:
                return b[p + 16:p + 16 + n]
            p += 16 + n + (-n % 8)
        raise KeyError("global heap object %d" % index)

    def _attr(self, body):
        ver, nlen, dlen, slen = struct.unpack_from("<BxHHH", body, 0)
        if ver == 1:
            pad = lambda n: n + (-n % 8)
            p = 8
        else:
            pad = lambda n: n
            p = 8 + (1 if ver == 3 else 0)
        name = body[p:p + nlen].split(b"\0")[0].decode()
        p += pad(nlen)
        dt = body[p:p + dlen]
        p += pad(dlen)
        shape = self._space(body[p:p + slen])
        p += pad(slen)
        kind, _ = self._dtype(dt)
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if isinstance(kind, np.dtype):
            a = np.frombuffer(body, dtype=kind, count=n, offset=p).reshape(shape)
            return name, (a.copy() if shape else a.reshape(())[()])
        if kind[0] == "str":
            return name, body[p:p + kind[1]].split(b"\0")[0].decode()
        if kind[0] == "vlen_str":
            vals = []
            for k in range(n):
                ln, addr, idx = struct.unpack_from("<IQI", body, p + 16 * k)
                vals.append(self._gheap(addr, idx)[:ln].decode() if ln else "")
            return name, (vals[0] if not shape else vals)
        return name, None


def read(path):
    return Reader(path)
