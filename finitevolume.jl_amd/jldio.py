"""Reader for the `mesh.jld` files of the reference's DFN workflow (examples/fractures/setupmesh.jl:46 writes them with
JLD.save, examples/fractures/ex.jl:9 reads them back): Julia's JLD 0.1 container is an HDF5 file behind a 512-byte
user block.  There is no HDF5 library in this environment, so this is a reader for the subset of the HDF5 file
format those files use — nothing else:

  * superblock version 0 / 1, 8-byte offsets and lengths, version-1 object headers (with continuation blocks);
  * groups stored as symbol tables (B-tree v1 + local heap), as link messages in the header, or as link messages in
    a fractal heap (root direct block, or the direct blocks of a root indirect block);
  * datasets with contiguous or compact layout, no filters;
  * element types: IEEE float64 / float32, fixed-point integers, and compounds of equal-sized members (how JLD writes
    `Array{Pair{Int,Int},1}`: the `neighbors` list), inline or as committed (shared) datatypes.

Anything outside that raises JLDFormatError instead of guessing.  Host-side pre-processing only.
"""

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"


class JLDFormatError(ValueError):
    pass


def _need(cond, what):
    if not cond:
        raise JLDFormatError("unsupported or damaged HDF5/JLD structure: " + what)


class _File:
    def __init__(self, data):
        self.b = data
        sb = 0
        while self.b[sb : sb + 8] != _SIG:  # the superblock sits at 0, 512, 1024, 2048, ...
            sb = 512 if sb == 0 else 2 * sb
            _need(sb + 8 <= len(self.b), "no HDF5 signature")
        version = self.b[sb + 8]
        _need(version in (0, 1), "superblock version %d" % version)
        self.O, self.L = self.b[sb + 13], self.b[sb + 14]
        _need(self.O == 8 and self.L == 8, "offset/length sizes %d/%d" % (self.O, self.L))
        p = sb + 24 + (4 if version == 1 else 0)
        self.base = self.u(p, 8)  # every address in the file is relative to this
        self.undefined = (1 << 64) - 1
        self.root_header = self.u(p + 4 * 8 + 8, 8)  # root symbol-table entry: link name offset, object header address

    def u(self, p, n):
        _need(p >= 0 and p + n <= len(self.b), "read past the end of the file")
        return int.from_bytes(self.b[p : p + n], "little")

    # -- object headers ---------------------------------------------------------------------------------------------
    def messages(self, header):
        """(type, flags, data position, size) of every message of a version-1 object header."""
        p = self.base + header
        _need(self.u(p, 1) == 1, "object header version %d" % self.u(p, 1))
        count, size = self.u(p + 2, 2), self.u(p + 8, 4)
        blocks, out = [(p + 16, size)], []
        while blocks and len(out) < count:
            q, left = blocks.pop(0)
            end = q + left
            while q + 8 <= end and len(out) < count:
                mtype, msize, flags = self.u(q, 2), self.u(q + 2, 2), self.u(q + 4, 1)
                if mtype == 0x10:  # continuation: more messages elsewhere
                    blocks.append((self.base + self.u(q + 8, 8), self.u(q + 16, 8)))
                out.append((mtype, flags, q + 8, msize))
                q += 8 + msize
        return out

    # -- groups -----------------------------------------------------------------------------------------------------
    def members(self, header):
        """name -> object header address of a group's members, or None when the object is not a group."""
        msgs = self.messages(header)
        table = [m for m in msgs if m[0] == 0x11]
        if table:
            return self._symbol_table(self.u(table[0][2], 8), self.u(table[0][2] + 8, 8))
        if not any(m[0] in (0x02, 0x06) for m in msgs):
            return None
        links = {}
        for mtype, _, d, _ in msgs:
            if mtype == 0x06:
                self._link(d, links)
            elif mtype == 0x02:  # link info: where the dense storage is
                heap = self.u(d + 2 + (8 if self.u(d + 1, 1) & 1 else 0), 8)
                if heap != self.undefined:
                    self._fractal_heap_links(heap, links)
        return links

    def _symbol_table(self, btree, heap):
        hp = self.base + heap
        _need(self.b[hp : hp + 4] == b"HEAP", "local heap signature")
        names = self.base + self.u(hp + 24, 8)
        out = {}

        def walk(node):
            p = self.base + node
            _need(self.b[p : p + 4] == b"TREE" and self.u(p + 4, 1) == 0, "group B-tree node")
            level, used = self.u(p + 5, 1), self.u(p + 6, 2)
            q = p + 24
            for _ in range(used):
                child = self.u(q + 8, 8)
                q += 16
                if level > 0:
                    walk(child)
                    continue
                s = self.base + child
                _need(self.b[s : s + 4] == b"SNOD", "symbol table node")
                for k in range(self.u(s + 6, 2)):
                    e = s + 8 + 40 * k
                    name = self.b[names + self.u(e, 8) :]
                    out[name[: name.index(b"\0")].decode()] = self.u(e + 8, 8)

        walk(btree)
        return out

    def _link(self, d, links):
        """One link message at d; returns the position behind it, or None when there is no message there."""
        if d + 4 > len(self.b) or self.u(d, 1) != 1:
            return None
        flags, q, kind = self.u(d + 1, 1), d + 2, 0
        if flags & 8:
            kind, q = self.u(q, 1), q + 1
        if flags & 4:
            q += 8  # creation order
        if flags & 16:
            q += 1  # character set
        width = 1 << (flags & 3)
        n, q = self.u(q, width), q + width
        name, q = self.b[q : q + n].decode(), q + n
        if kind == 0:  # hard link
            links[name] = self.u(q, 8)
            return q + 8
        return q + 2 + self.u(q, 2)  # soft / external link: skipped

    def _fractal_heap_links(self, heap, links):
        p = self.base + heap
        _need(self.b[p : p + 4] == b"FRHP" and self.u(p + 4, 1) == 0, "fractal heap header")
        _need(self.u(p + 7, 2) == 0, "filtered fractal heap")
        flags = self.u(p + 9, 1)
        q = p + 14 + 8 + 8 + 8 + 8 + 8 * 8  # to the doubling-table parameters
        width, start, max_direct, max_bits = self.u(q, 2), self.u(q + 2, 8), self.u(q + 10, 8), self.u(q + 18, 2)
        root, rows = self.u(q + 22, 8), self.u(q + 30, 2)
        offset_bytes = (max_bits + 7) // 8

        def direct(addr, size):
            s = self.base + addr
            _need(self.b[s : s + 4] == b"FHDB", "fractal heap direct block")
            at, end = s + 5 + 8 + offset_bytes + (4 if flags & 2 else 0), s + size
            while at is not None and at < end - 4:
                at = self._link(at, links)

        if rows == 0:
            direct(root, start)
            return
        s = self.base + root
        _need(self.b[s : s + 4] == b"FHIB", "fractal heap indirect block")
        at = s + 5 + 8 + offset_bytes
        for r in range(rows):
            size = start if r < 2 else start << (r - 1)
            _need(size <= max_direct, "fractal heap with nested indirect blocks")
            for _ in range(width):
                child, at = self.u(at, 8), at + 8
                if child != self.undefined:
                    direct(child, size)

    # -- datasets ---------------------------------------------------------------------------------------------------
    def _dtype(self, d):
        cls, size = self.u(d, 1) & 15, self.u(d + 4, 4)
        if cls == 0:
            return np.dtype("<%s%d" % ("i" if self.u(d + 1, 1) & 8 else "u", size)), 1
        if cls == 1:
            _need(size in (4, 8), "%d-byte float" % size)
            return np.dtype("<f%d" % size), 1
        if cls == 6:  # compound: read as rows of its members when they all have the first member's type
            version, members = self.u(d, 1) >> 4, self.u(d + 1, 2)
            q, kinds = d + 8, []
            for _ in range(members):
                end = self.b.index(b"\0", q)
                if version < 3:
                    q = q + ((end - q) // 8 + 1) * 8  # name padded to a multiple of 8
                    q += 4 if version == 2 else 4 + 1 + 3 + 4 + 4 + 16
                else:
                    q = end + 1 + max(1, ((size).bit_length() + 7) // 8)
                kind, _ = self._dtype(q)
                kinds.append(kind)
                q += 8 + (4 if self.u(q, 1) & 15 == 0 else 12)  # fixed-point: 4 property bytes; float: 12
            _need(all(k == kinds[0] for k in kinds) and kinds[0].itemsize * members == size, "compound with mixed members")
            return kinds[0], members
        raise JLDFormatError("datatype class %d (only numbers and compounds of numbers are read)" % cls)

    def dataset(self, header):
        dims = dtype = raw = None
        for mtype, flags, d, _ in self.messages(header):
            if mtype == 0x01:  # dataspace
                version, rank = self.u(d, 1), self.u(d + 1, 1)
                at = d + (8 if version == 1 else 4)
                dims = [self.u(at + 8 * i, 8) for i in range(rank)]
            elif mtype == 0x03:  # datatype, inline or committed
                if flags & 2:
                    shared = self.u(d + (8 if self.u(d, 1) == 1 else 2), 8)
                    inner = [m for m in self.messages(shared) if m[0] == 0x03]
                    _need(inner, "committed datatype without a datatype message")
                    dtype = self._dtype(inner[0][2])
                else:
                    dtype = self._dtype(d)
            elif mtype == 0x08:  # layout
                _need(self.u(d, 1) == 3, "data layout version %d" % self.u(d, 1))
                kind = self.u(d + 1, 1)
                if kind == 0:
                    raw = self.b[d + 4 : d + 4 + self.u(d + 2, 2)]
                elif kind == 1:
                    addr, size = self.u(d + 2, 8), self.u(d + 10, 8)
                    raw = b"" if addr == self.undefined else self.b[self.base + addr : self.base + addr + size]
                else:
                    raise JLDFormatError("chunked dataset")
            elif mtype == 0x0B:
                raise JLDFormatError("filtered (compressed) dataset")
        _need(dims is not None and dtype is not None and raw is not None, "dataset without dataspace, datatype or layout")
        kind, members = dtype
        count = int(np.prod(dims)) if dims else 1
        _need(len(raw) >= count * members * kind.itemsize, "dataset shorter than its dataspace")
        a = np.frombuffer(raw, kind, count * members)
        if members > 1:
            _need(len(dims) <= 1, "array of compounds with more than one dimension")
            return a.reshape(count, members).copy()
        # HDF5 is row-major, Julia column-major: JLD writes the dimensions reversed, so the transpose is Julia's array
        return a.reshape(dims).T.copy()


def load_jld(path, *names):
    """JLD.load(path, names...): the named top-level variables as numpy arrays (all of them, as a dict, without names).
    `Array{Pair{Int,Int},1}` comes back as an (n, 2) integer array."""
    with open(path, "rb") as f:
        h5 = _File(f.read())
    top = h5.members(h5.root_header)
    _need(top is not None, "root object is not a group")
    if not names:
        out = {}
        for name, header in top.items():
            if name.startswith("_") or h5.members(header) is not None:
                continue  # JLD's own bookkeeping (_creator, _types, _refs)
            out[name] = h5.dataset(header)
        return out
    missing = [n for n in names if n not in top]
    if missing:
        raise KeyError("no variable %s in %s (has: %s)" % (", ".join(missing), path, ", ".join(sorted(k for k in top if not k.startswith("_")))))
    values = tuple(h5.dataset(top[n]) for n in names)
    return values[0] if len(values) == 1 else values
