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
    return dims, data_sid, (b"ZIP" in desc[:k]), desc[k2 + 8:]


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
