"""Minimal reader for the HDF5 files the reference ships with the hot path (no h5py/libhdf5 in this environment).

Handles exactly the subset those files use (verified byte-wise on robot2d.h5 and the ten autopilot gain files):
superblock version 0, version-1 object headers (with continuation blocks), groups stored either as version-1 B-trees +
symbol-table nodes + local heaps or as Link messages in the group's object header, dataspace versions 1/2, fixed-point and IEEE floating-point datatypes, data layout version 3
(contiguous or compact). Datasets come back as numpy arrays in JULIA orientation: HDF5.jl writes a column-major
Julia array with its dimensions reversed, so the flat data is reshaped in Fortran order with the dims flipped.

Reference call sites that read these files: lib/FlightApps/src/robot2d/robot2d.jl:419-421,
lib/FlightPhysics/src/control.jl:879-935.
"""
from __future__ import annotations

import struct
import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5Error(ValueError):
    pass


class File:
    def __init__(self, path: str):
        with open(path, "rb") as f:
            self.b = f.read()
        b = self.b
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise HDF5Error("not an HDF5 file")
        if b[8] != 0:
            raise HDF5Error(f"superblock version {b[8]} not supported (only 0)")
        if b[13] != 8 or b[14] != 8:
            raise HDF5Error("only 8-byte offsets/lengths supported")
        self.base = struct.unpack_from("<Q", b, 24)[0]
        # root group symbol table entry starts at 24 + 4*8 = 56
        ent = 56
        self.root_header = struct.unpack_from("<Q", b, ent + 8)[0]
        cache_type = struct.unpack_from("<I", b, ent + 16)[0]
        if cache_type == 1:
            self.root_btree, self.root_heap = struct.unpack_from("<QQ", b, ent + 24)
        else:
            self.root_btree, self.root_heap = self._group_addrs(self.root_header)
        self._names = None

    # ---- object headers -------------------------------------------------------------------------
    def _messages(self, addr: int):
        b = self.b
        ver, _, nmsg, _refc, hsize = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise HDF5Error(f"object header version {ver} not supported")
        blocks = [(addr + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                data = pos + 8
                if mtype == 0x0010:  # continuation
                    off, ln = struct.unpack_from("<QQ", b, data)
                    blocks.append((off + self.base, ln))
                out.append((mtype, data, msize))
                pos = data + msize
        return out

    def _group_addrs(self, header: int):
        for mtype, data, _ in self._messages(header):
            if mtype == 0x0011:  # symbol table message
                return struct.unpack_from("<QQ", self.b, data)
        raise HDF5Error("object is not an old-style group")

    # ---- group traversal ------------------------------------------------------------------------
    def _heap_data(self, heap: int) -> int:
        if self.b[heap:heap + 4] != b"HEAP":
            raise HDF5Error("bad local heap")
        return struct.unpack_from("<Q", self.b, heap + 24)[0] + self.base

    def _walk_btree(self, node: int, heap_data: int, out: dict):
        b = self.b
        if b[node:node + 4] == b"SNOD":
            n = struct.unpack_from("<H", b, node + 6)[0]
            for k in range(n):
                e = node + 8 + 40 * k
                name_off, hdr = struct.unpack_from("<QQ", b, e)
                end = b.index(b"\x00", heap_data + name_off)
                out[b[heap_data + name_off:end].decode("utf-8")] = hdr + self.base
            return
        if b[node:node + 4] != b"TREE":
            raise HDF5Error("bad group B-tree node")
        used = struct.unpack_from("<H", b, node + 6)[0]
        p = node + 24 + 8  # skip header (24) and key 0
        for _ in range(used):
            child = struct.unpack_from("<Q", b, p)[0]
            self._walk_btree(child + self.base, heap_data, out)
            p += 16  # child pointer + next key

    def _link_messages(self, header: int, out: dict):
        """Version-1 Link messages (type 0x0006) stored directly in a group's object header ("compact" groups,
        what HDF5.jl / libhdf5 >= 1.8 writes for small groups)."""
        b = self.b
        for mtype, data, _ in self._messages(header):
            if mtype != 0x0006:
                continue
            ver, flags = b[data], b[data + 1]
            if ver != 1:
                raise HDF5Error(f"link message version {ver} not supported")
            p = data + 2
            ltype = 0
            if flags & 0x08:
                ltype = b[p]; p += 1
            if flags & 0x04:
                p += 8  # creation order
            if flags & 0x10:
                p += 1  # character set
            nlen_size = 1 << (flags & 0x03)
            nlen = int.from_bytes(b[p:p + nlen_size], "little"); p += nlen_size
            name = b[p:p + nlen].decode("utf-8"); p += nlen
            if ltype != 0:
                continue  # soft / external links: not used by the reference's files
            out[name] = struct.unpack_from("<Q", b, p)[0] + self.base

    def _children(self, header: int) -> dict:
        """name -> object header address of the members of the group whose object header is at `header`."""
        out: dict = {}
        for mtype, data, _ in self._messages(header):
            if mtype == 0x0011:  # old-style group: B-tree + local heap
                btree, heap = struct.unpack_from("<QQ", self.b, data)
                if btree != UNDEF and heap != UNDEF:
                    self._walk_btree(btree + self.base, self._heap_data(heap + self.base), out)
        self._link_messages(header, out)
        return out

    def _is_group(self, header: int) -> bool:
        types = {m[0] for m in self._messages(header)}
        return 0x0001 not in types   # no dataspace message: not a dataset

    def names(self) -> dict:
        """Flat map 'group/sub/dataset' -> object header address of every dataset in the file."""
        if self._names is None:
            out: dict = {}

            def walk(header, prefix):
                for name, addr in self._children(header).items():
                    if self._is_group(addr):
                        walk(addr, prefix + name + "/")
                    else:
                        out[prefix + name] = addr
            root: dict = {}
            if self.root_btree != UNDEF and self.root_heap != UNDEF:
                self._walk_btree(self.root_btree + self.base, self._heap_data(self.root_heap + self.base), root)
            self._link_messages(self.root_header + self.base, root)
            for name, addr in root.items():
                if self._is_group(addr):
                    walk(addr, name + "/")
                else:
                    out[name] = addr
            self._names = out
        return self._names

    def keys(self):
        return list(self.names().keys())

    # ---- datasets -------------------------------------------------------------------------------
    def read(self, name: str) -> np.ndarray:
        hdr = self.names()[name]
        b = self.b
        dims = None
        dtype = None
        raw = None
        for mtype, data, msize in self._messages(hdr):
            if mtype == 0x0001:  # dataspace
                ver, rank = b[data], b[data + 1]
                start = data + (8 if ver == 1 else 4)
                dims = struct.unpack_from(f"<{rank}Q", b, start) if rank else ()
            elif mtype == 0x0003:  # datatype
                cls = b[data] & 0x0F
                bits0 = b[data + 1]
                size = struct.unpack_from("<I", b, data + 4)[0]
                endian = ">" if (bits0 & 1) else "<"
                if cls == 1:
                    dtype = np.dtype(f"{endian}f{size}")
                elif cls == 0:
                    signed = (bits0 >> 3) & 1
                    dtype = np.dtype(f"{endian}{'i' if signed else 'u'}{size}")
                else:
                    raise HDF5Error(f"datatype class {cls} not supported")
            elif mtype == 0x0008:  # data layout
                ver, lclass = b[data], b[data + 1]
                if ver != 3:
                    raise HDF5Error(f"layout version {ver} not supported")
                if lclass == 1:
                    addr, size = struct.unpack_from("<QQ", b, data + 2)
                    if addr == UNDEF:
                        raw = b""
                    else:
                        raw = b[addr + self.base: addr + self.base + size]
                elif lclass == 0:
                    size = struct.unpack_from("<H", b, data + 2)[0]
                    raw = b[data + 4: data + 4 + size]
                else:
                    raise HDF5Error("chunked layout not supported")
        if dims is None or dtype is None or raw is None:
            raise HDF5Error(f"dataset {name}: incomplete header")
        count = int(np.prod(dims)) if dims else 1
        arr = np.frombuffer(raw, dtype=dtype, count=count).astype(dtype.newbyteorder("="))
        if not dims:
            return arr.reshape(())
        return arr.reshape(tuple(reversed(dims)), order="F")  # Julia orientation


def read_all(path: str) -> dict:
    f = File(path)
    return {k: f.read(k) for k in f.keys()}
