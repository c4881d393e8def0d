"""Genotype sources for the host-side mirror of the reference's R API.

The reference reads genotypes through gdsfmt (not part of the reference tree,
SURVEY.md section 1, L0).  Its block reader (``CdSNPWorkSpace::snpRead``,
src/dGenGWAS.cpp:677-733) is *kept* in a real R deployment;
this module only provides what the Python host mirror and the tests need:

* :class:`GenoFile` -- an in-memory SNP GDS look-alike (``sample.id``,
  ``snp.id``, ``snp.chromosome``, ``genotype``) with the same node names that
  ``snpgdsOpen`` documents (R/AllUtilities.R:32-155);
* :func:`open_gds` -- a minimal reader for *uncompressed-bit2* SNP GDS files as
  written by SNPRelate (enough for ``inst/extdata/hapmap_geno.gds``, committed
  as the fixture ``tests/golden/hapmap_geno.gds``).  Container format: 12-byte
  magic, blocks with 6-byte little-endian size/next pointers, one stream per
  node.

Genotypes are held 2-bit packed, SNP-major ("sample.order": samples fastest
inside a SNP), 4 genotypes per byte LSB first, value 3 = missing -- the layout
of the GDS ``genotype`` node -- with each SNP row padded to a whole byte.
"""
import struct
import zlib

import numpy as np

_MAGIC = b"COREARRAYx0A"


def pack_2bit_rows(geno_u8):
    """uint8 [L][N] (values 0..3) -> uint8 [L][ceil(N/4)] 2-bit packed rows."""
    g = np.asarray(geno_u8, dtype=np.uint8)
    g = np.minimum(g, 3)
    L, N = g.shape
    nb = (N + 3) // 4
    pad = nb * 4 - N
    if pad:
        g = np.concatenate([g, np.full((L, pad), 3, np.uint8)], axis=1)
    g = g.reshape(L, nb, 4)
    return (g[:, :, 0] | (g[:, :, 1] << 2) | (g[:, :, 2] << 4) | (g[:, :, 3] << 6)).astype(np.uint8)


def unpack_2bit_rows(packed, n_samp):
    """uint8 [L][ceil(N/4)] -> uint8 [L][N]."""
    p = np.asarray(packed, dtype=np.uint8)
    L = p.shape[0]
    out = np.empty((L, p.shape[1], 4), np.uint8)
    for k in range(4):
        out[:, :, k] = (p >> (2 * k)) & 3
    return out.reshape(L, -1)[:, :n_samp]


class GenoFile:
    """In-memory SNP GDS look-alike.

    ``genotype`` may be given as uint8 [n_snp][n_samp] (0/1/2, >2 missing) or
    through ``packed`` (uint8 [n_snp][ceil(n_samp/4)]).
    """

    def __init__(self, genotype=None, sample_id=None, snp_id=None, snp_chromosome=None,
                 packed=None, n_samp=None):
        if packed is None:
            genotype = np.asarray(genotype, dtype=np.uint8)
            n_snp, n_samp = genotype.shape
            packed = pack_2bit_rows(genotype)
        else:
            packed = np.ascontiguousarray(packed, dtype=np.uint8)
            n_snp = packed.shape[0]
            assert n_samp is not None and packed.shape[1] == (n_samp + 3) // 4
        self.packed = packed
        self.n_snp, self.n_samp = int(n_snp), int(n_samp)
        self.sample_id = (np.arange(1, n_samp + 1) if sample_id is None
                          else np.asarray(sample_id))
        self.snp_id = (np.arange(1, n_snp + 1, dtype=np.int32) if snp_id is None
                       else np.asarray(snp_id))
        self.snp_chromosome = (np.ones(n_snp, np.int32) if snp_chromosome is None
                               else np.asarray(snp_chromosome))
        # snpgdsOption() defaults (R/AllUtilities.R:1910-1991)
        self.autosome_start, self.autosome_end = 1, 22
        assert len(self.sample_id) == self.n_samp and len(self.snp_id) == self.n_snp

    def read_genotype(self, snp_sel=None, samp_sel=None):
        """uint8 [n_sel_snp][n_sel_samp], like snpRead with selections."""
        p = self.packed if snp_sel is None else self.packed[np.asarray(snp_sel)]
        g = unpack_2bit_rows(p, self.n_samp)
        if samp_sel is not None:
            g = g[:, np.asarray(samp_sel)]
        return np.ascontiguousarray(g)


# ---------------------------------------------------------------------------
# minimal GDS container reader
# ---------------------------------------------------------------------------
def _u48(b):
    return int.from_bytes(b, "little")


def _read_streams(d):
    if d[:12] != _MAGIC:
        raise ValueError("not a GDS file (bad magic)")
    pos, blocks = 18, {}
    heads = {}
    while pos < len(d):
        sz = _u48(d[pos:pos + 6])
        head = (sz >> 47) & 1
        sz &= (1 << 47) - 1
        nxt = _u48(d[pos + 6:pos + 12])
        if sz < 12:
            raise ValueError("corrupt GDS block")
        if head:
            sid = struct.unpack("<I", d[pos + 12:pos + 16])[0]
            slen = _u48(d[pos + 16:pos + 22])
            heads[sid] = (pos, slen)
            blocks[pos] = (pos + 22, pos + sz, nxt)
        else:
            blocks[pos] = (pos + 12, pos + sz, nxt)
        pos += sz
    streams = {}
    for sid, (pos, slen) in heads.items():
        parts, p = [], pos
        while True:
            a, b, nxt = blocks[p]
            parts.append(d[a:b])
            if not nxt:
                break
            p = nxt
        streams[sid] = b"".join(parts)[:slen]
    return streams


def _dir_entries(s):
    """(name, descriptor stream id) pairs of a folder stream."""
    out, i = [], 0
    while True:
        i = s.find(b"\xf5\x00", i)
        if i < 0:
            break
        sid = struct.unpack("<I", s[i + 2:i + 6])[0]
        j = s.find(b"\x16\x44\xc6\x60\x10", i)
        if j < 0 or j - i > 40:
            i += 2
            continue
        ln = s[j + 5]
        out.append((s[j + 6:j + 6 + ln].decode("latin1"), sid))
        i = j + 6 + ln
    return out


def _node_info(desc):
    """(dims, data stream id, is_zip, attribute blob) of an array descriptor."""
    k = desc.find(b"\xc3\x43\x61")
    nb = desc[k + 3]
    dims = list(struct.unpack("<%dI" % (nb // 4), desc[k + 4:k + 4 + nb]))
    k2 = desc.find(b"\xc4\xc3\x7c\x0c")
    data_sid = struct.unpack("<I", desc[k2 + 4:k2 + 8])[0]
    coder = _node_coder(desc[:k])
    if coder not in ("", "ZIP"):
        raise ValueError("GDS node compressed with %s: this reader handles uncompressed nodes and plain zlib streams (\"ZIP\") only "
                         "-- gdsfmt's LZ4 / LZMA coders and the random-access containers (*_RA) are not restated here" % coder)
    return dims, data_sid, coder == "ZIP", desc[k2 + 8:]


def _node_coder(head):
    """Compression coder named in an array descriptor (the bytes before its dimensions): "" (none), "ZIP", "ZIP_RA", "LZ4",
    "LZ4_RA", "LZMA", "LZMA_RA".  gdsfmt stores the coder as a tagged, length-prefixed text property (tag c4 46 6d 10, one
    length byte, the name with its level / block size after a colon); the name is read AT that position -- a chance "ZIP" or
    "LZ4" among the class-name or attribute bytes of an uncompressed node names no coder."""
    k = head.find(b"\xc4\x46\x6d\x10")
    if k < 0 or k + 5 > len(head):
        return ""
    n = head[k + 4]
    # "ZIP.max:..." / "ZIP.fast": a compression LEVEL after a dot is still the plain coder; "_RA" is part of the name
    text = head[k + 5:k + 5 + n].split(b":")[0].split(b".")[0].decode("latin1").upper()
    return text


def open_gds(path):
    """Read a SNPRelate-written SNP GDS file with an uncompressed bit2
    ``genotype`` node into a :class:`GenoFile`."""
    with open(path, "rb") as f:
        d = f.read()
    streams = _read_streams(d)
    root_sid = struct.unpack("<I", d[14:18])[0]
    entries = dict(_dir_entries(streams[root_sid]))
    for need in ("sample.id", "snp.id", "snp.chromosome", "genotype"):
        if need not in entries:
            raise ValueError("GDS node '%s' not found" % need)

    def raw(name):
        dims, sid, is_zip, attr = _node_info(streams[entries[name]])
        b = streams[sid]
        if is_zip:
            b = zlib.decompress(b)
        return dims, b, attr

    dims, b, _ = raw("sample.id")
    sample_id = np.array([x.decode("latin1") for x in b.split(b"\x00")[:dims[0]]])
    dims, b, _ = raw("snp.id")
    snp_id = np.frombuffer(b, "<i4", count=dims[0]).copy()
    dims, b, _ = raw("snp.chromosome")
    if len(b) == dims[0]:
        chrom = np.frombuffer(b, np.uint8, count=dims[0]).astype(np.int32)
    else:
        chrom = np.frombuffer(b, "<i4", count=dims[0]).copy()
    dims, b, attr = raw("genotype")
    if b"ZIP" in streams[entries["genotype"]][:64] and len(b) * 4 < dims[0] * dims[1]:
        raise ValueError("compressed genotype nodes are not supported by this reader")
    total = dims[0] * dims[1]
    bits = np.frombuffer(b, np.uint8)
    g = np.empty((len(bits), 4), np.uint8)
    for k in range(4):
        g[:, k] = (bits >> (2 * k)) & 3
    g = g.reshape(-1)[:total].reshape(dims[0], dims[1])
    if b"sample.order" in attr:
        # dims = [n_snp][n_samp] (samples fastest): RDim_Sample_X_SNP,
        # src/dGenGWAS.cpp:576-589
        geno = g
    else:
        # snp.order: dims = [n_samp][n_snp]
        geno = np.ascontiguousarray(g.T)
    return GenoFile(genotype=geno, sample_id=sample_id, snp_id=snp_id, snp_chromosome=chrom)


# ---------------------------------------------------------------------------
# Streaming block reader: the `genotype` node straight into 2-bit rows, block by block
# ---------------------------------------------------------------------------
# What CdSNPWorkSpace::snpRead + CGenoReadBySNP do in the reference (src/dGenGWAS.cpp:677-733, :1218-1397: a sequential
# block iterator over the selected SNPs with a one-block prefetch) -- minus the inflation to one byte per genotype: the
# node's own 2 bits per genotype go to the device as they are (SNPGPU_GENO_PACKED2), realigned so that every SNP row
# starts on a byte (the node packs the bit stream continuously: with 279 samples a row is 69.75 bytes long).  The file
# is never loaded as a whole: block headers are walked with seeks, the data stream is read (and, for zlib-compressed
# nodes, inflated) incrementally.
class _ByteStream:
    """Sequential byte ranges [lo, hi) of a GDS data stream given as file extents [(offset, length)], optionally
    zlib-compressed.  Ranges must not move backwards by more than the bytes still buffered (the reader re-reads at
    most the last byte of the previous block)."""

    def __init__(self, fobj, extents, length, zipped=False, chunk=1 << 20):
        self.f, self.extents, self.length, self.chunk = fobj, list(extents), int(length), int(chunk)
        self.z = zlib.decompressobj() if zipped else None
        self.ext_i, self.ext_off = 0, 0          # next raw byte to fetch
        self.raw_left = self.length if not zipped else None
        self.buf, self.pos0 = bytearray(), 0     # inflated / raw bytes [pos0, pos0 + len(buf))

    def _fetch_raw(self, want):
        out = bytearray()
        while want > 0 and self.ext_i < len(self.extents):
            off, ln = self.extents[self.ext_i]
            take = min(want, ln - self.ext_off)
            if take > 0:
                self.f.seek(off + self.ext_off)
                out += self.f.read(take)
                self.ext_off += take
                want -= take
            if self.ext_off >= ln:
                self.ext_i, self.ext_off = self.ext_i + 1, 0
        return bytes(out)

    def _fill(self, upto):
        while self.pos0 + len(self.buf) < upto:
            raw = self._fetch_raw(self.chunk)
            if not raw:
                raise ValueError("GDS genotype stream ends early")
            self.buf += self.z.decompress(raw) if self.z is not None else raw

    def read(self, lo, hi):
        if lo < self.pos0:
            raise ValueError("GDS stream reader cannot seek backwards")
        self._fill(hi)
        a = lo - self.pos0
        out = bytes(self.buf[a:a + (hi - lo)])
        if a > 0:                                  # drop what is behind `lo`
            del self.buf[:a]
            self.pos0 = lo
        return out


def realign_bit2_rows(seg, seg_bit0, n_samp, row_begin, row_end, out=None):
    """Rows [row_begin, row_end) of a continuous 2-bit stream (row r starts at bit 2 * n_samp * r of the stream) as
    byte-aligned rows uint8 [rows][ceil(n_samp / 4)], samples beyond n_samp in the last byte set to 3 (missing).
    `seg`: bytes covering the rows, its first byte holding stream bit `seg_bit0` (a multiple of 8)."""
    n_rows, rb = row_end - row_begin, (n_samp + 3) // 4
    if out is None:
        out = np.empty((n_rows, rb), np.uint8)
    a = np.frombuffer(bytes(seg) + b"\x00\x00", np.uint8)
    period = {0: 1, 4: 2}.get((2 * n_samp) % 8, 4)              # rows r, r + period start with the same bit shift
    stride = 2 * n_samp * period // 8
    for c in range(min(period, n_rows)):
        bit = 2 * n_samp * (row_begin + c) - seg_bit0
        o, sh = bit >> 3, bit & 7
        cnt = (n_rows - c + period - 1) // period
        v = np.lib.stride_tricks.as_strided(a[o:], shape=(cnt, rb + 1), strides=(stride, 1), writeable=False)
        dst = out[c::period]
        if sh == 0:
            dst[:] = v[:, :rb]
        else:
            dst[:] = (v[:, :rb] >> sh) | ((v[:, 1:] << (8 - sh)) & 0xFF)
    tail = rb * 4 - n_samp
    if tail:
        out[:, -1] |= (0xFF << (2 * (4 - tail))) & 0xFF
    return out


class GenoStream:
    """Streaming view of a SNP GDS file: metadata in memory, `genotype` on disk.

        gs = open_gds_stream(path)
        for snp_begin, n_snp, rows in gs.blocks(16384):      # rows: uint8 [n_snp][ceil(n_samp / 4)] (SNPGPU_GENO_PACKED2)
            acc.feed(rows, fmt=GENO_PACKED2)

    `blocks(block_snps, buffers=[b0, b1])` fills caller-provided (page-locked) buffers in turn instead of allocating."""

    def __init__(self, path, sample_id, snp_id, snp_chromosome, dims, extents, length, zipped, sample_order):
        self.path, self.sample_id, self.snp_id, self.snp_chromosome = path, sample_id, snp_id, snp_chromosome
        self._dims, self._extents, self._length, self._zipped = dims, extents, length, zipped
        self.sample_order = sample_order
        self.n_snp, self.n_samp = (dims[0], dims[1]) if sample_order else (dims[1], dims[0])
        self.autosome_start, self.autosome_end = 1, 22

    def blocks(self, block_snps, buffers=None, snp_begin=0, snp_end=None):
        snp_end = self.n_snp if snp_end is None else min(int(snp_end), self.n_snp)
        n, rb = self.n_samp, (self.n_samp + 3) // 4
        if not self.sample_order:
            yield from self._blocks_snp_order(block_snps, buffers, snp_begin, snp_end)
            return
        with open(self.path, "rb") as f:
            st = _ByteStream(f, self._extents, self._length, self._zipped)
            turn = 0
            for lo in range(snp_begin, snp_end, block_snps):
                hi = min(lo + block_snps, snp_end)
                b0, b1 = (2 * n * lo) >> 3, (2 * n * hi + 7) >> 3
                seg = st.read(b0, b1)
                out = None
                if buffers is not None:
                    out = np.asarray(buffers[turn % len(buffers)]).reshape(-1)[: (hi - lo) * rb].reshape(hi - lo, rb)
                    turn += 1
                yield lo, hi - lo, realign_bit2_rows(seg, 8 * b0, n, lo, hi, out)

    def _blocks_snp_order(self, block_snps, buffers, snp_begin, snp_end):
        # dims = [n_samp][n_snp], SNPs fastest: a block of SNPs is a strided read over every sample's row
        # (snpRead transposes as well, src/dGenGWAS.cpp:699-731).  A raw single-extent node is read through a memory map; a
        # zlib-compressed or chained one cannot be addressed by sample row, so it is inflated ONCE into host memory (2 n L / 8
        # bytes; refused above SNPGPU_GDS_INFLATE_MAX bytes, default 8 GiB -- convert such a file to sample.order, the layout
        # snpgdsBED2GDS writes by default, or leave it uncompressed)
        n, L, rb = self.n_samp, self.n_snp, (self.n_samp + 3) // 4
        if self._zipped or len(self._extents) != 1:
            mm = getattr(self, "_inflated", None)
            if mm is None:
                import os
                need = (2 * n * L + 7) // 8
                cap = int(os.environ.get("SNPGPU_GDS_INFLATE_MAX", 8 << 30))
                if need > cap:
                    raise ValueError("a compressed snp.order genotype node of %d bytes does not fit the in-memory budget of %d bytes "
                                     "(SNPGPU_GDS_INFLATE_MAX)" % (need, cap))
                with open(self.path, "rb") as f:
                    mm = np.frombuffer(_ByteStream(f, self._extents, self._length, self._zipped).read(0, need), np.uint8)
                self._inflated = mm
        else:
            mm = np.memmap(self.path, dtype=np.uint8, mode="r", offset=self._extents[0][0], shape=(self._extents[0][1],))
        turn = 0
        for lo in range(snp_begin, snp_end, block_snps):
            hi = min(lo + block_snps, snp_end)
            g = np.empty((n, hi - lo), np.uint8)
            for i in range(n):
                bit0 = 2 * (i * L + lo)
                seg = np.asarray(mm[bit0 >> 3: ((2 * (i * L + hi) + 7) >> 3)])
                codes = np.stack([(seg >> (2 * k)) & 3 for k in range(4)], 1).reshape(-1)
                g[i] = codes[(bit0 & 7) >> 1:][: hi - lo]
            rows = pack_2bit_rows(np.ascontiguousarray(g.T))
            if buffers is not None:
                out = np.asarray(buffers[turn % len(buffers)]).reshape(-1)[: (hi - lo) * rb].reshape(hi - lo, rb)
                out[:] = rows
                rows = out
                turn += 1
            yield lo, hi - lo, rows

    def read_packed(self, block_snps=65536):
        """the whole genotype node as 2-bit rows (small files; tests)"""
        return np.concatenate([r.copy() for _, _, r in self.blocks(block_snps)], 0)

    # the in-memory interface of GenoFile, for the host mirror of the R functions (api.py works on a working space that
    # holds the selected genotypes, as the reference's .InitFile2 + gnrSetGenoSpace do): materialised on first use
    @property
    def packed(self):
        if getattr(self, "_packed", None) is None:
            self._packed = self.read_packed()
        return self._packed

    def read_genotype(self, snp_sel=None, samp_sel=None):
        p = self.packed if snp_sel is None else self.packed[np.asarray(snp_sel)]
        g = unpack_2bit_rows(p, self.n_samp)
        if samp_sel is not None:
            g = g[:, np.asarray(samp_sel)]
        return np.ascontiguousarray(g)


def _walk_blocks(f):
    """Block table of a GDS file by seeking: {first block offset: (stream id, stream length)} and the chain of every
    block -- without reading any payload."""
    f.seek(0, 2)
    size = f.tell()
    f.seek(0)
    if f.read(12) != _MAGIC:
        raise ValueError("not a GDS file (bad magic)")
    pos, blocks, heads = 18, {}, {}
    while pos < size:
        f.seek(pos)
        h = f.read(22)
        sz = _u48(h[0:6])
        head = (sz >> 47) & 1
        sz &= (1 << 47) - 1
        nxt = _u48(h[6:12])
        if sz < 12:
            raise ValueError("corrupt GDS block")
        if head:
            heads[struct.unpack("<I", h[12:16])[0]] = (pos, _u48(h[16:22]))
            blocks[pos] = (pos + 22, pos + sz, nxt)
        else:
            blocks[pos] = (pos + 12, pos + sz, nxt)
        pos += sz
    return blocks, heads


def open_gds_stream(path):
    """Open a SNP GDS file for block-wise reading (GenoStream): only the small nodes are read into memory."""
    with open(path, "rb") as f:
        blocks, heads = _walk_blocks(f)
        f.seek(14)
        root_sid = struct.unpack("<I", f.read(4))[0]

        def extents(sid):
            pos, slen = heads[sid]
            ext, left = [], slen
            while True:
                a, b, nxt = blocks[pos]
                take = min(b - a, left)
                if take > 0:
                    ext.append((a, take))
                left -= take
                if not nxt or left <= 0:
                    break
                pos = nxt
            return ext, slen

        def load(sid):
            ext, slen = extents(sid)
            out = bytearray()
            for a, ln in ext:
                f.seek(a)
                out += f.read(ln)
            return bytes(out)

        entries = dict(_dir_entries(load(root_sid)))
        for need in ("sample.id", "snp.id", "snp.chromosome", "genotype"):
            if need not in entries:
                raise ValueError("GDS node '%s' not found" % need)

        def small(name):
            dims, sid, is_zip, attr = _node_info(load(entries[name]))
            b = load(sid)
            return dims, (zlib.decompress(b) if is_zip else b)

        dims, b = small("sample.id")
        sample_id = np.array([x.decode("latin1") for x in b.split(b"\x00")[:dims[0]]])
        dims, b = small("snp.id")
        snp_id = np.frombuffer(b, "<i4", count=dims[0]).copy()
        dims, b = small("snp.chromosome")
        chrom = (np.frombuffer(b, np.uint8, count=dims[0]).astype(np.int32) if len(b) == dims[0]
                 else np.frombuffer(b, "<i4", count=dims[0]).copy())
        gdesc = load(entries["genotype"])
        dims, sid, is_zip, attr = _node_info(gdesc)
        ext, slen = extents(sid)
        if is_zip:
            f.seek(ext[0][0])
            if f.read(1) != b"\x78":         # (a coder tag this reader does not know would already have raised in _node_info)
                raise ValueError("the genotype node's data do not start a zlib stream; this reader inflates plain zlib streams only")
        elif slen != (2 * dims[0] * dims[1] + 7) // 8:      # an uncompressed bit2 node is exactly its genotypes, no more, no less
            raise ValueError("the genotype node holds %d bytes where %d x %d 2-bit genotypes need %d: compressed or damaged data"
                             % (slen, dims[0], dims[1], (2 * dims[0] * dims[1] + 7) // 8))
    return GenoStream(path, sample_id, snp_id, chrom, dims, ext, slen, is_zip, b"sample.order" in attr)


# ---------------------------------------------------------------------------
# SNPRELATE_OUTPUT files (snpgdsGRM(out.fn=), snpgdsMergeGRM)
# ---------------------------------------------------------------------------
def write_output(path, nodes):
    """Store the nodes of a ``SNPRELATE_OUTPUT`` file (R/IBD.R:567-586: root attribute ``FileFormat``, nodes
    ``command``, ``sample.id``, ``snp.id``, ``grm`` and, for IndivBeta, ``avg_val``).  In an R deployment gdsfmt
    writes these as a GDS file and is kept; the Python mirror stores the same nodes in a numpy archive under the
    exact file name given."""
    arrays = {"FileFormat": np.array("SNPRELATE_OUTPUT")}
    for k, v in nodes.items():
        arrays[k] = np.asarray(v)
    with open(path, "wb") as f:
        np.savez(f, **arrays)


def read_output(path):
    with open(path, "rb") as f:
        if f.read(9) == b"COREARRAY":
            raise ValueError("'%s' is a GDS file written by gdsfmt; this Python mirror reads only the numpy archives its "
                             "own snpgdsGRM(out_fn=) / snpgdsPCACorr(outgds=) write (same nodes, different container)" % path)
    with np.load(path, allow_pickle=False) as z:
        nodes = {k: z[k] for k in z.files}
    if "grm" in nodes and nodes["grm"].dtype != np.float64:
        nodes["grm"] = nodes["grm"].astype(np.float64)        # out.prec = "single"
    if str(nodes.get("FileFormat")) != "SNPRELATE_OUTPUT":
        raise ValueError("'%s' is not valid." % path)          # R/IBD.R:659
    return nodes
