"""A small HDF5 reader / writer for the `.spy` container's data files (io/load_spy_container.py:34,
io/save_spy_container.py:19): NumPy + struct only - h5py is not part of the GPU image.

What a Syncopy data file is (save_spy_container.py:196-222): one HDF5 file whose root group holds the dataset
"data" (contiguous, written in one piece by `create_dataset(name, data=...)`), the dataset "trialdefinition"
(created with maxshape=(None, ncol), hence CHUNKED) and a few root attributes mirroring the JSON side-car.

Reader: superblock versions 0-3, object headers of version 1 and 2, groups as symbol tables (B-tree v1 + local heap)
or as compact link messages, fixed-point / floating-point / two-member compound (complex) types, little endian,
layouts compact / contiguous / chunked-with-B-tree-v1 without filters.  Anything else raises HDF5FormatError - it
never guesses.  Contiguous datasets are returned as np.memmap (the container's documented raw access, `data_offset`
in the .info file), so an AnalogData file of any size is staged to the GPU without an extra host copy.

Writer: superblock 0, one root group (symbol table), contiguous datasets, fixed-length string / numeric root
attributes: the subset libhdf5 of any version (and therefore unmodified Syncopy through h5py) reads back.
Layout follows the HDF5 File Format Specification version 1.1/2.0 (III.A-E, IV.A.1-2); no code of the reference or
of libhdf5 is involved.
"""
import mmap
import os
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5FormatError(Exception):
    pass


# a version-1 object header message carries its size as uint16: one attribute holds < 64 KiB (HDF5 itself has the same
# limit for attributes kept in the object header; h5py then raises RuntimeError, save_spy_container.py:263-272)
MAX_ATTRIBUTE_BYTES = 65528


class AttributeTooLarge(HDF5FormatError):
    def __init__(self, name, size):
        super().__init__(f"attribute '{name}' needs {size} bytes, more than the {MAX_ATTRIBUTE_BYTES} an object header "
                         f"message holds")
        self.name, self.size = name, size


# ======================================================================================================== reader
class _Reader:
    def __init__(self, path):
        # the metadata is a few KB anywhere in the file: map it read-only instead of reading a multi-GB recording
        # into host memory (slicing, .index and np.frombuffer work on an mmap); close() drops the mapping once the
        # datasets are resolved - contiguous datasets come back as their own np.memmap
        self.path = path
        self._fh = open(path, "rb")
        size = os.fstat(self._fh.fileno()).st_size
        if size == 0:
            self._fh.close()
            raise HDF5FormatError(f"{path}: empty file")
        self.buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        self.base = 0
        try:
            self._superblock()
        except Exception:
            self.close()
            raise

    def close(self):
        if getattr(self, "buf", None) is not None:
            try:
                self.buf.close()
            except BufferError:      # an array built with np.frombuffer still points into the mapping: leave it to the GC
                pass
            self.buf = None
        if getattr(self, "_fh", None) is not None:
            self._fh.close()
            self._fh = None

    # ---- primitives
    def u(self, off, size):
        return int.from_bytes(self.buf[off:off + size], "little")

    def _superblock(self):
        b = self.buf
        off = 0
        while off < len(b) and b[off:off + 8] != SIGNATURE:      # the signature may sit at 0, 512, 1024, ...
            off = 512 if off == 0 else off * 2
        if off >= len(b):
            raise HDF5FormatError(f"{self.path}: not an HDF5 file (signature not found)")
        ver = b[off + 8]
        if ver in (0, 1):
            self.O, self.L = b[off + 13], b[off + 14]
            p = off + 24 + (4 if ver == 1 else 0)
            self.base = self.u(p, self.O)
            p += 4 * self.O                                        # base, free space, end of file, driver info
            self.root = self.u(p + self.O, self.O)                 # root symbol table entry: name offset, header
        elif ver in (2, 3):
            self.O, self.L = b[off + 9], b[off + 10]
            self.base = self.u(off + 12, self.O)
            self.root = self.u(off + 12 + 3 * self.O, self.O)
        else:
            raise HDF5FormatError(f"{self.path}: superblock version {ver} is not supported")
        if self.O != 8 or self.L != 8:
            raise HDF5FormatError(f"{self.path}: offsets/lengths of {self.O}/{self.L} bytes are not supported")

    # ---- object headers
    def messages(self, addr):
        """[(type, flags, payload offset, payload size)] of the object header at `addr`, continuations followed."""
        a = self.base + addr
        out = []
        if self.buf[a:a + 4] == b"OHDR":
            flags = self.buf[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            csz = 1 << (flags & 3)
            size0 = self.u(p, csz)
            p += csz
            blocks = [(p, size0)]
            track = bool(flags & 0x04)
            while blocks:
                p, n = blocks.pop(0)
                end = p + n
                while p + 4 <= end:
                    mtype, msize, mflags = self.buf[p], self.u(p + 1, 2), self.buf[p + 3]
                    p += 4 + (2 if track else 0)
                    if mtype == 0x10:
                        ca, cl = self.u(p, 8), self.u(p + 8, 8)
                        blocks.append((self.base + ca + 4, cl - 8))            # "OCHK" + ... + checksum
                    elif mtype != 0:
                        out.append((mtype, mflags, p, msize))
                    p += msize
            return out
        if self.buf[a] != 1:
            raise HDF5FormatError(f"{self.path}: object header at {addr} has unknown version {self.buf[a]}")
        nmsg, hsize = self.u(a + 2, 2), self.u(a + 8, 4)
        blocks = [(a + 16, hsize)]
        while blocks and len(out) < nmsg + 64:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end:
                mtype, msize, mflags = self.u(p, 2), self.u(p + 2, 2), self.buf[p + 4]
                p += 8
                if mtype == 0x10:
                    blocks.append((self.base + self.u(p, 8), self.u(p + 8, 8)))
                elif mtype != 0:
                    out.append((mtype, mflags, p, msize))
                p += msize
        return out

    # ---- groups
    def links(self, addr):
        """{name: object header address} of the group whose header is at `addr`."""
        out = {}
        for mtype, _, p, n in self.messages(addr):
            if mtype == 0x11:                                       # symbol table: B-tree + local heap
                btree, heap = self.u(p, 8), self.u(p + 8, 8)
                h = self.base + heap
                if self.buf[h:h + 4] != b"HEAP":
                    raise HDF5FormatError(f"{self.path}: local heap signature missing")
                seg = self.base + self.u(h + 24, 8)
                self._group_node(btree, seg, out)
            elif mtype == 0x06:                                     # link message (compact new-style group)
                ver, fl = self.buf[p], self.buf[p + 1]
                q = p + 2
                ltype = 0
                if fl & 0x08:
                    ltype = self.buf[q]
                    q += 1
                if fl & 0x04:
                    q += 8
                if fl & 0x10:
                    q += 1
                lsz = 1 << (fl & 3)
                ln = self.u(q, lsz)
                q += lsz
                name = self.buf[q:q + ln].decode("utf-8")
                q += ln
                if ltype == 0:
                    out[name] = self.u(q, 8)
            elif mtype == 0x02:
                # link info: dense storage (fractal heap) if the heap address is defined
                fl = self.buf[p + 1]
                q = p + 2 + (8 if fl & 1 else 0)
                if self.u(q, 8) != UNDEF:
                    raise HDF5FormatError(f"{self.path}: densely stored groups (fractal heap) are not supported")
        return out

    def _group_node(self, addr, seg, out):
        a = self.base + addr
        if self.buf[a:a + 4] == b"SNOD":
            n = self.u(a + 6, 2)
            for k in range(n):
                e = a + 8 + 40 * k
                noff, hdr = self.u(e, 8), self.u(e + 8, 8)
                end = self.buf.find(b"\0", seg + noff)
                if end < 0:
                    raise HDF5FormatError(f"{self.path}: unterminated link name in the local heap (truncated file?)")
                out[self.buf[seg + noff:end].decode("utf-8")] = hdr
            return
        if self.buf[a:a + 4] != b"TREE" or self.buf[a + 4] != 0:
            raise HDF5FormatError(f"{self.path}: group B-tree node expected at {addr}")
        n = self.u(a + 6, 2)
        p = a + 24
        for k in range(n):
            child = self.u(p + 8 + 16 * k, 8)                       # key, child, key, child, ..., key
            self._group_node(child, seg, out)

    # ---- datasets
    def _dtype(self, p):
        cls, ver = self.buf[p] & 0x0F, self.buf[p] >> 4
        bits = self.buf[p + 1:p + 4]
        size = self.u(p + 4, 4)
        if cls in (0, 1) and bits[0] & 1:
            raise HDF5FormatError(f"{self.path}: big-endian data are not supported")
        if cls == 0:
            return np.dtype(("<i" if bits[0] & 0x08 else "<u") + str(size)), 8 + 4
        if cls == 1:
            if size not in (2, 4, 8):
                raise HDF5FormatError(f"{self.path}: {size}-byte floats are not supported")
            return np.dtype("<f" + str(size)), 8 + 12
        if cls == 6:
            nmemb = bits[0] | (bits[1] << 8)
            q = p + 8
            names, offs, types = [], [], []
            for _ in range(nmemb):
                end = self.buf.find(b"\0", q)
                if end < 0:
                    raise HDF5FormatError(f"{self.path}: unterminated member name in a compound datatype (truncated file?)")
                name = self.buf[q:end].decode("ascii")
                if ver < 3:
                    q += (end - q + 8) // 8 * 8
                else:
                    q = end + 1
                if ver == 1:
                    off = self.u(q, 4)
                    q += 4 + 1 + 3 + 4 + 4 + 16
                elif ver == 2:
                    off = self.u(q, 4)
                    q += 4
                else:
                    nb = 1 if size < 256 else 2 if size < 65536 else 4
                    off = self.u(q, nb)
                    q += nb
                t, used = self._dtype(q)
                q += used
                names.append(name)
                offs.append(off)
                types.append(t)
            if names == ["r", "i"] and types[0] == types[1] and types[0].kind == "f" and offs == [0, types[0].itemsize] \
                    and size == 2 * types[0].itemsize:
                return np.dtype("<c" + str(size)), q - p
            return np.dtype({"names": names, "formats": types, "offsets": offs, "itemsize": size}), q - p
        raise HDF5FormatError(f"{self.path}: datatype class {cls} is not supported")

    def dataset(self, addr):
        shape = dtype = layout = None
        filtered = False
        for mtype, _, p, n in self.messages(addr):
            if mtype == 0x01:
                ver, rank, fl = self.buf[p], self.buf[p + 1], self.buf[p + 2]
                q = p + (8 if ver == 1 else 4)
                shape = tuple(self.u(q + 8 * k, 8) for k in range(rank))
            elif mtype == 0x03:
                dtype, _ = self._dtype(p)
            elif mtype == 0x0B:
                filtered = True
            elif mtype == 0x08:
                ver = self.buf[p]
                if ver != 3:
                    raise HDF5FormatError(f"{self.path}: data layout message version {ver} is not supported")
                cls = self.buf[p + 1]
                if cls == 0:
                    layout = ("compact", p + 4, self.u(p + 2, 2))
                elif cls == 1:
                    layout = ("contiguous", self.u(p + 2, 8), self.u(p + 10, 8))
                elif cls == 2:
                    nd = self.buf[p + 2]
                    layout = ("chunked", self.u(p + 3, 8), tuple(self.u(p + 11 + 4 * k, 4) for k in range(nd)))
                else:
                    raise HDF5FormatError(f"{self.path}: layout class {cls} is not supported")
        if shape is None or dtype is None or layout is None:
            raise HDF5FormatError(f"{self.path}: object at {addr} is not a dataset")
        if filtered:
            raise HDF5FormatError(f"{self.path}: filtered (compressed) datasets are not supported")
        count = int(np.prod(shape)) if shape else 1
        if layout[0] == "compact":
            return np.frombuffer(self.buf, dtype, count, layout[1]).reshape(shape).copy(), None
        if layout[0] == "contiguous":
            if layout[1] == UNDEF or count == 0:
                return np.zeros(shape, dtype), None
            off = self.base + layout[1]
            return np.memmap(self.path, dtype=dtype, mode="r", offset=off, shape=shape, order="C"), off
        out = np.zeros(shape, dtype)
        cdims = layout[2][:-1]
        if layout[1] != UNDEF:
            self._chunk_node(layout[1], len(shape), cdims, out)
        return out, None

    def _chunk_node(self, addr, rank, cdims, out):
        a = self.base + addr
        if self.buf[a:a + 4] != b"TREE" or self.buf[a + 4] != 1:
            raise HDF5FormatError(f"{self.path}: chunk B-tree node expected at {addr}")
        level, n = self.buf[a + 5], self.u(a + 6, 2)
        ksz = 8 + 8 * (rank + 1)
        p = a + 24
        for k in range(n):
            key = p + k * (ksz + 8)
            nbytes, mask = self.u(key, 4), self.u(key + 4, 4)
            offs = tuple(self.u(key + 8 + 8 * d, 8) for d in range(rank))
            child = self.u(key + ksz, 8)
            if level > 0:
                self._chunk_node(child, rank, cdims, out)
                continue
            if mask:
                raise HDF5FormatError(f"{self.path}: filtered chunks are not supported")
            chunk = np.frombuffer(self.buf, out.dtype, int(np.prod(cdims)), self.base + child).reshape(cdims)
            sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, out.shape))
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]


def read_datasets(path, names=None):
    """{name: (array, file offset of the first element or None)} of the datasets in the root group of `path`.
    Contiguous datasets come back as read-only np.memmap."""
    r = _Reader(path)
    try:
        links = r.links(r.root)
        out = {}
        for name, addr in links.items():
            if names is not None and name not in names:
                continue
            out[name] = r.dataset(addr)
        if names is not None:
            for n in names:
                if n not in out:
                    raise KeyError(f"{path}: no dataset named '{n}' in the root group (has: {sorted(links)})")
        return out
    finally:
        r.close()


# ======================================================================================================== writer
GROUP_LEAF_K, GROUP_INTERNAL_K = 4, 16
DATA_ALIGN = 4096          # raw data start on page boundaries (memmap / O_DIRECT friendly; any 8-byte multiple is legal)


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


def _float_type(size):
    if size == 4:
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 31, 0, 4, 0, 32, 23, 8, 0, 23, 127)
    if size == 8:
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 63, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
    if size == 2:
        return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 15, 0, 2, 0, 16, 10, 5, 0, 10, 15)
    raise HDF5FormatError(f"{size}-byte floats cannot be written")


def _type_message(dt):
    dt = np.dtype(dt)
    if dt.byteorder == ">":
        raise HDF5FormatError("big-endian arrays cannot be written")
    if dt.kind == "f":
        return _float_type(dt.itemsize)
    if dt.kind in "iu":
        return struct.pack("<BBBBIHH", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    if dt.kind == "b":
        return struct.pack("<BBBBIHH", 0x10, 0x08, 0, 0, 1, 0, 8)
    if dt.kind == "c":                                      # h5py's convention: compound {r, i}
        half = dt.itemsize // 2
        body = b""
        for k, name in enumerate((b"r", b"i")):
            body += _pad8(name + b"\0") + struct.pack("<IB3xII4I", k * half, 0, 0, 0, 0, 0, 0, 0) + _float_type(half)
        return struct.pack("<BBBBI", 0x16, 2, 0, 0, dt.itemsize) + body
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)
    raise HDF5FormatError(f"dtype {dt} cannot be written")


def _space_message(shape):
    if len(shape) == 0:
        return struct.pack("<BBBB4x", 1, 0, 0, 0)
    return struct.pack("<BBBB4x", 1, len(shape), 1, 0) + struct.pack(f"<{2 * len(shape)}Q", *shape, *shape)


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(messages, min_size=0):
    body = b"".join(messages)
    n = len(messages)
    if len(body) < min_size:                                 # a NIL message fills the rest
        body += struct.pack("<HHB3x", 0, min_size - len(body) - 8, 0) + b"\0" * (min_size - len(body) - 8)
        n += 1
    return struct.pack("<BBHII4x", 1, 0, n, 1, len(body)) + body


def _attribute(name, value):
    if isinstance(value, str):
        value = np.bytes_(value.encode("utf-8"))
    elif isinstance(value, (list, tuple)) and all(isinstance(v, str) for v in value):
        value = np.array([v.encode("utf-8") for v in value] or [b""])
    arr = np.asarray(value)
    if arr.dtype.kind == "U":
        arr = np.char.encode(arr, "utf-8")
    if arr.dtype.kind == "S" and arr.dtype.itemsize == 0:
        arr = arr.astype("S1")
    if arr.dtype.kind == "O":
        raise HDF5FormatError(f"attribute '{name}': object arrays cannot be written")
    nm = name.encode("utf-8") + b"\0"
    dt, sp = _type_message(arr.dtype), _space_message(arr.shape)
    body = struct.pack("<BBHHH", 1, 0, len(nm), len(dt), len(sp)) + _pad8(nm) + _pad8(dt) + _pad8(sp)
    raw = np.ascontiguousarray(arr).tobytes()
    if len(body) + len(raw) + 8 > MAX_ATTRIBUTE_BYTES:
        raise AttributeTooLarge(name, len(body) + len(raw))
    return _message(0x0C, body + raw)


def write_file(path, datasets, attrs=None):
    """Write `datasets` {name: array} (contiguous layout, C order) and root attributes `attrs` {name: number, str,
    list of str or numeric array} into a new HDF5 file.  Returns {name: file offset of the dataset's first element}
    (what h5py's Dataset.id.get_offset() reports, save_spy_container.py:228-231)."""
    names = sorted(datasets, key=lambda s: s.encode("utf-8"))
    if not names or len(names) > 2 * GROUP_LEAF_K:
        raise HDF5FormatError(f"between 1 and {2 * GROUP_LEAF_K} datasets per file, got {len(names)}")
    arrays = {}
    for n in names:
        a = datasets[n]
        if not isinstance(a, np.ndarray):
            a = np.asarray(a)
        if not a.flags["C_CONTIGUOUS"]:
            a = np.ascontiguousarray(a)
        arrays[n] = a
    # ---- local heap: "" at 0 (the root's own name and key 0), the names, one free block at the end
    seg = bytearray(8)
    name_off = {}
    for n in names:
        name_off[n] = len(seg)
        seg += _pad8(n.encode("utf-8") + b"\0")
    free_at = len(seg)
    seg += struct.pack("<QQ", 1, 32) + b"\0" * 16                                  # next = 1 (none), size 32
    # ---- addresses
    root_msgs_tail = [_attribute(k, v) for k, v in (attrs or {}).items()]
    root_size = 16 + 24 + sum(len(m) for m in root_msgs_tail)
    a_root = 96
    a_btree = a_root + root_size
    a_heap = a_btree + 24 + (2 * GROUP_INTERNAL_K) * 8 + (2 * GROUP_INTERNAL_K + 1) * 8
    a_seg = a_heap + 32
    a_snod = a_seg + len(seg)
    a_hdr = a_snod + 8 + 2 * GROUP_LEAF_K * 40
    hdr_addr, hdrs = {}, {}
    probe = a_hdr
    def dataset_header(a, addr):
        msgs = [_message(0x01, _space_message(a.shape)),
                _message(0x03, _type_message(a.dtype), flags=1),
                _message(0x05, struct.pack("<BBBBI", 2, 2, 2, 1, 0), flags=1),
                _message(0x08, struct.pack("<BBQQ", 3, 1, addr, a.nbytes))]
        body = sum(len(m) for m in msgs)
        return _object_header(msgs, min_size=256 if body + 8 <= 256 else 0)

    for n in names:                       # header sizes do not depend on the data addresses: first pass sizes them
        hdr_addr[n] = probe
        probe += len(dataset_header(arrays[n], 0))
    data_addr = {}
    pos = -(-probe // DATA_ALIGN) * DATA_ALIGN
    for n in names:
        data_addr[n] = pos if arrays[n].nbytes else UNDEF
        pos = -(-(pos + arrays[n].nbytes) // DATA_ALIGN) * DATA_ALIGN if arrays[n].nbytes else pos
    eof = max([probe] + [data_addr[n] + arrays[n].nbytes for n in names if arrays[n].nbytes])
    for n in names:
        hdrs[n] = dataset_header(arrays[n], data_addr[n])
    # ---- assemble the metadata block
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, GROUP_LEAF_K, GROUP_INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, a_root, 1, 0) + struct.pack("<QQ", a_btree, a_heap)
    assert len(sb) == 96
    root = _object_header([_message(0x11, struct.pack("<QQ", a_btree, a_heap))] + root_msgs_tail)
    assert len(root) == root_size
    bt = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1, UNDEF, UNDEF)
    bt += struct.pack("<QQQ", 0, a_snod, name_off[names[-1]])
    bt += b"\0" * (a_heap - a_btree - len(bt))
    heap = b"HEAP" + struct.pack("<B3xQQQ", 0, len(seg), free_at, a_seg)
    snod = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
    for n in names:
        snod += struct.pack("<QQII16x", name_off[n], hdr_addr[n], 0, 0)
    snod += b"\0" * (8 + 2 * GROUP_LEAF_K * 40 - len(snod))
    meta = sb + root + bt + heap + bytes(seg) + snod + b"".join(hdrs[n] for n in names)
    assert len(meta) == probe
    with open(path, "wb") as fh:
        fh.write(meta)
        for n in names:
            a = arrays[n]
            if not a.nbytes:
                continue
            fh.write(b"\0" * (data_addr[n] - fh.tell()))
            fh.write(memoryview(a.reshape(-1).view(np.uint8)))
    return {n: (None if data_addr[n] == UNDEF else data_addr[n]) for n in names}
