#!/usr/bin/env python3
"""Generate the committed fixtures under tests/golden/ from the reference's own
test data (run once in the build container; /root/reference is NOT available on
the GPU box, so tests only ever read the files written here).

Inputs (data files only, no source code):
  /root/reference/inst/extdata/hapmap_geno.gds           (example data set, copied verbatim)
  /root/reference/inst/unitTests/valid/Validate.IBS.RData   -> validate_ibs.npz
  /root/reference/inst/unitTests/valid/Validate.KING.RData  -> validate_king.npz
  /root/reference/inst/unitTests/valid/Validate.PCA.RData   -> validate_pca.npz
  .../Validate.MoM.RData, Validate.Beta.RData, Validate.EIGMIX.RData -> validate_mom/beta/eigmix.npz
  (test.PLINK.MoM :193-224, test.IndivBeta :277-304, test.EIGMIX :308-327)
These are the golden vectors of inst/unitTests/test_rel.R (test.IBS :97-124,
test.KING :228-273, test.PCA :128-142).

The .RData files are xz-compressed R serialisation format 2 ("RDX2", XDR);
a small parser for the handful of SEXP types they contain is below.
"""
import lzma
import os
import shutil
import struct
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class RData:
    def __init__(self, raw):
        assert raw[:5] == b"RDX2\n" and raw[5:7] == b"X\n", raw[:8]
        self.b, self.p, self.refs = raw, 7, []
        self.i32(); self.i32(); self.i32()

    def i32(self):
        v = struct.unpack(">i", self.b[self.p:self.p + 4])[0]
        self.p += 4
        return v

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
        if t == 254:                       # NULL
            return None
        if t == 255:                       # reference
            return self.refs[(flags >> 8) - 1]
        if t == 1:                         # symbol
            s = self.item()
            self.refs.append(s)
            return s
        if t == 9:                         # CHARSXP
            n = self.i32()
            if n < 0:
                return None
            s = self.b[self.p:self.p + n].decode("latin1")
            self.p += n
            return s
        if t == 2:                         # pairlist
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                out.append((tag, self.item()))
                flags = self.i32()
                t = flags & 0xFF
                has_attr, has_tag = bool(flags & 0x200), bool(flags & 0x400)
                if t == 254:
                    return out
                assert t == 2, t
        if t in (10, 13):                  # logical / integer
            n = self.i32()
            v = np.frombuffer(self.b, ">i4", n, self.p).astype(np.int32)
            self.p += 4 * n
        elif t == 14:                      # double
            n = self.i32()
            v = np.frombuffer(self.b, ">f8", n, self.p).astype(np.float64)
            self.p += 8 * n
        elif t == 16:                      # character vector
            n = self.i32()
            v = [self.item() for _ in range(n)]
        elif t == 19:                      # list
            n = self.i32()
            v = [self.item() for _ in range(n)]
        else:
            raise NotImplementedError("SEXP type %d" % t)
        if has_attr:
            attrs = dict(self.item())
            if "names" in attrs and isinstance(v, list) and t == 19:
                v = dict(zip(attrs["names"], v))
            elif "dim" in attrs and isinstance(v, np.ndarray):
                v = v.reshape(tuple(int(x) for x in attrs["dim"]), order="F")
        return v


def load_rdata(path):
    with open(path, "rb") as f:
        raw = lzma.decompress(f.read())
    return dict(RData(raw).item())


def main():
    shutil.copyfile(os.path.join(REF, "inst/extdata/hapmap_geno.gds"),
                    os.path.join(HERE, "hapmap_geno.gds"))
    vdir = os.path.join(REF, "inst/unitTests/valid")

    ibs = load_rdata(os.path.join(vdir, "Validate.IBS.RData"))["ibs"]
    np.savez_compressed(os.path.join(HERE, "validate_ibs.npz"),
                        ibs=ibs["ibs"], snp_id=ibs["snp.id"],
                        sample_id=np.array(ibs["sample.id"]))

    king = load_rdata(os.path.join(vdir, "Validate.KING.RData"))[".king"]
    v1, v2 = king
    np.savez_compressed(os.path.join(HERE, "validate_king.npz"),
                        robust_IBS0=v1["IBS0"], robust_kinship=v1["kinship"],
                        homo_k0=v2["k0"], homo_k1=v2["k1"],
                        snp_id=v1["snp.id"], snp_id_homo=v2["snp.id"],
                        sample_id=np.array(v1["sample.id"]))

    pca = load_rdata(os.path.join(vdir, "Validate.PCA.RData"))[".rv"]
    np.savez_compressed(os.path.join(HERE, "validate_pca.npz"), genmat=pca["genmat"], corr=pca["corr"],
                        snploading=pca["snploading"], samploading=pca["samploading"])
    mom = load_rdata(os.path.join(vdir, "Validate.MoM.RData"))["ibd"]
    np.savez_compressed(os.path.join(HERE, "validate_mom.npz"), k0=mom["k0"], k1=mom["k1"],
                        afreq=mom["afreq"], snp_id=mom["snp.id"])
    beta = load_rdata(os.path.join(vdir, "Validate.Beta.RData"))[".beta"]
    np.savez_compressed(os.path.join(HERE, "validate_beta.npz"), beta=beta["beta"], snp_id=beta["snp.id"])
    eigmix = load_rdata(os.path.join(vdir, "Validate.EIGMIX.RData"))[".eigmix"]
    np.savez_compressed(os.path.join(HERE, "validate_eigmix.npz"), ibd=eigmix)
    for k in ("validate_ibs", "validate_king", "validate_pca", "validate_mom", "validate_beta", "validate_eigmix"):
        z = np.load(os.path.join(HERE, k + ".npz"))
        print(k, {n: z[n].shape for n in z.files})


if __name__ == "__main__":
    sys.exit(main())
