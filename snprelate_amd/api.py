"""Host-side mirror of the reference's R interface for the pairwise hot path.

Same function names, argument names/meaning, defaults, return fields and error
behaviour as the R functions (R is absent from the build image, so the host
side above the C ABI is Python; the R shim a maintainer would add is in
INTEGRATION.md):

    snpgdsOpen / snpgdsClose      R/AllUtilities.R:32-155   (in-memory GenoFile)
    snpgdsIBS, snpgdsIBSNum       R/IBS.R:22-73
    snpgdsIBDKING                 R/IBD.R:333-419
    snpgdsGRM                     R/IBD.R:543-615  (methods GCTA, Eigenstrat, Corr, EIGMIX/Weighted, IndivBeta)
    snpgdsMergeGRM                R/IBD.R:624-741
    snpgdsIBDMoM                  R/IBD.R:22-68    (PLINK method of moments)
    snpgdsIndivBeta               R/IBD.R:838-866
    snpgdsEIGMIX                  R/PCA.R:311-338
    snpgdsPCA                     R/PCA.R:22-91    (algorithm="exact")
    snpgdsSNPRateFreq             R/AllUtilities.R (allele freq / MAF / missing rate)

All arithmetic runs on the MI355X through libsnpgpu.so (`_lib`); there is no
CPU fallback.  R's ``NULL`` is ``None``, ``NaN`` is ``float('nan')``; R lists
are dicts with the same field names.
"""
import ctypes
import math

import numpy as np

from . import _lib
from . import gds as _gds
from .gds import GenoFile, open_gds, pack_2bit_rows  # noqa: F401




def snpgdsOpen(filename, stream=False, **_):
    """snpgdsOpen (R/AllUtilities.R:32-155).  stream=True: the genotype node stays on disk and is read block by block
    (gds.GenoStream: `.blocks(n)` feeds accumulators with 2-bit rows; the snpgds* functions below still work on it -- they set
    up a working space of the selected genotypes, for which the node is read once on first use)."""
    if stream:
        from .gds import open_gds_stream
        return open_gds_stream(filename)
    return open_gds(filename)


def snpgdsClose(gdsobj):
    return None


def _cat(verbose, *a):
    if verbose:
        print(*a, sep="")


def _init_file2(cmd, gdsobj, sample_id, snp_id, autosome_only=True, remove_monosnp=True,
                maf=float("nan"), missing_rate=float("nan"), num_thread=1, verbose=True, device=0, allele_freq=None):
    """.InitFile2, R/Internal.R:166-484: sample/SNP selection, autosome filter,
    gnrSetGenoSpace, gnrSelSNP_Base (gnrSelSNP_Base_Ex when allele.freq is given), gnrGetGenoDim.
    allele_freq follows the reference's bookkeeping: given per entry of `snp_id` (or per SNP of the file), it is
    brought into DATASET order with match(snp.ids[kept], snp.id) (R/Internal.R:355,370,405), drives the SNP filter
    (non-finite = excluded) and is returned subset to the surviving SNPs as ws["allele_freq"]."""
    if not isinstance(gdsobj, GenoFile):
        raise TypeError("'gdsobj' should be a SNP GDS object (snpgdsOpen / GenoFile)")
    if num_thread is None or (isinstance(num_thread, float) and math.isnan(num_thread)):
        import os
        num_thread = os.cpu_count() or 1
    num_thread = int(num_thread)
    if num_thread < 1:
        raise ValueError("`num.thread' should be a positive value or NA.")
    _cat(verbose and cmd, cmd)

    sample_ids = gdsobj.sample_id
    samp_flag = None
    if sample_id is not None:
        want = np.asarray(sample_id)
        samp_flag = np.isin(sample_ids, want)
        if int(samp_flag.sum()) != len(want):
            raise ValueError("Some of sample.id do not exist!")
        if samp_flag.sum() <= 0:
            raise ValueError("No sample in the working dataset.")
        sample_ids = sample_ids[samp_flag]

    snp_ids = gdsobj.snp_id
    snp_flag = np.ones(len(snp_ids), bool)
    want = None
    if allele_freq is not None:
        allele_freq = np.ascontiguousarray(allele_freq, np.float64)
    if snp_id is not None:
        want = np.asarray(snp_id)
        if allele_freq is not None and len(allele_freq) != len(want):
            raise ValueError("'length(allele.freq)' should be 'length(snp.id)'.")
        snp_flag = np.isin(snp_ids, want)
        if int(snp_flag.sum()) != len(want):
            raise ValueError("Some of snp.id do not exist!")
        if snp_flag.sum() <= 0:
            raise ValueError("No SNP in the working dataset.")
    elif allele_freq is not None and len(allele_freq) != len(snp_ids):
        raise ValueError("'length(allele.freq)' should be the number of SNPs.")
    if autosome_only is not False:
        chrom = gdsobj.snp_chromosome
        if autosome_only is True:
            # gnrChromRangeNumeric, src/SNPRelate.cpp:1035-1062 with snpgdsOption() defaults
            auto = (chrom >= gdsobj.autosome_start) & (chrom <= gdsobj.autosome_end)
            m = int(len(snp_ids) - (snp_flag & auto).sum())
            _cat(verbose, "Excluding %d SNP%s (non-autosomes or non-selection)" % (m, "" if m == 1 else "s"))
        else:
            auto = (chrom == autosome_only)
            _cat(verbose, "Keeping %d SNPs according to chromosome %s" % (int((snp_flag & auto).sum()), autosome_only))
        snp_flag &= auto
    if allele_freq is not None:
        if want is not None:
            # allele.freq[match(snp.ids[snp.id], tmp.id)]: position of every kept dataset SNP in the caller's list
            order = np.argsort(want, kind="stable")
            pos = order[np.searchsorted(want[order], snp_ids[snp_flag])]
            allele_freq = allele_freq[pos]
        else:
            allele_freq = allele_freq[snp_flag]
    snp_ids = snp_ids[snp_flag]

    # gnrSetGenoSpace: the selected rectangle becomes the working space
    packed = gdsobj.packed[snp_flag]
    n_samp = gdsobj.n_samp
    if samp_flag is not None and not samp_flag.all():
        from .gds import unpack_2bit_rows
        g = unpack_2bit_rows(packed, n_samp)[:, samp_flag]
        packed = pack_2bit_rows(g)
        n_samp = int(samp_flag.sum())
    packed = np.ascontiguousarray(packed)
    L = _lib.lib()
    _lib.check(L.snpgpu_ws_set_geno(_lib._ptr(packed), packed.shape[0], n_samp, _lib.GENO_PACKED2, int(device)))

    if remove_monosnp or np.isfinite(maf) or np.isfinite(missing_rate):
        t_maf, t_miss = maf, missing_rate
        if not np.isfinite(maf):
            maf = -1.0               # R/Internal.R:438-439
        if not np.isfinite(missing_rate):
            missing_rate = 2.0
        sel = np.zeros(packed.shape[0], np.uint8)
        nex = ctypes.c_int32(0)
        if allele_freq is None:
            _lib.check(L.snpgpu_ws_sel_snp_base(int(bool(remove_monosnp)), float(maf), float(missing_rate),
                                                ctypes.byref(nex), _lib._ptr(sel)))
        else:
            allele_freq = np.ascontiguousarray(allele_freq)
            _lib.check(L.snpgpu_ws_sel_snp_base_ex(_lib._ptr(allele_freq), int(bool(remove_monosnp)), float(maf),
                                                   float(missing_rate), ctypes.byref(nex), _lib._ptr(sel)))
            allele_freq = np.ascontiguousarray(allele_freq[sel.astype(bool)])
        snp_ids = snp_ids[sel.astype(bool)]
        packed = packed[sel.astype(bool)]
        _cat(verbose, "Excluding %d SNP%s (monomorphic: %s, MAF: %s, missing rate: %s)" %
             (nex.value, "" if nex.value == 1 else "s", str(bool(remove_monosnp)).upper(), t_maf, t_miss))

    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(L.snpgpu_ws_get_geno_dim(ctypes.byref(a), ctypes.byref(b)))
    if verbose:
        print("    # of samples: %d" % b.value)
        print("    # of SNPs: %d" % a.value)
        print("    using %d thread%s (the GPU path ignores num.thread)" % (num_thread, "" if num_thread == 1 else "s"))
    return dict(sample_id=sample_ids, snp_id=snp_ids, n_snp=a.value, n_samp=b.value,
                num_thread=num_thread, verbose=verbose, packed=packed, device=int(device), allele_freq=allele_freq)


def _tri_or_full(n, use_matrix):
    return np.empty(_lib.tri_size(n) if use_matrix else (n, n), np.float64)


def snpgdsSNPRateFreq(gdsobj, sample_id=None, snp_id=None, with_id=False, device=0):
    """Allele frequency, MAF and missing rate per SNP (Get_AF_MR_perSNP,
    src/dGenGWAS.cpp:472-552) over the selected samples."""
    ws = _init_file2(None, gdsobj, sample_id, snp_id, autosome_only=False, remove_monosnp=False,
                     verbose=False, device=device)
    L = ws["n_snp"]
    af, maf, mr = (np.empty(L, np.float64) for _ in range(3))
    _lib.check(_lib.lib().snpgpu_ws_snp_rate_freq(_lib._ptr(af), _lib._ptr(maf), _lib._ptr(mr)))
    rv = dict(AlleleFreq=af, MinorFreq=maf, MissingRate=mr)
    if with_id:
        rv.update(sample_id=ws["sample_id"], snp_id=ws["snp_id"])
    return rv


def snpgdsIBS(gdsobj, sample_id=None, snp_id=None, autosome_only=True, remove_monosnp=True,
              maf=float("nan"), missing_rate=0.01, num_thread=1, useMatrix=False, verbose=True, device=0):
    ws = _init_file2("Identity-By-State (IBS) analysis on genotypes:", gdsobj, sample_id, snp_id,
                     autosome_only, remove_monosnp, maf, missing_rate, num_thread, verbose, device)
    if not isinstance(useMatrix, bool):
        raise TypeError("is.logical(useMatrix) is not TRUE")
    out = _tri_or_full(ws["n_samp"], useMatrix)
    _lib.check(_lib.lib().snpgpu_gnrIBSAve(ws["num_thread"], int(useMatrix), int(verbose), _lib._ptr(out)))
    return dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], ibs=out)


def snpgdsIBSNum(gdsobj, sample_id=None, snp_id=None, autosome_only=True, remove_monosnp=True,
                 maf=float("nan"), missing_rate=0.01, num_thread=1, verbose=True, device=0):
    ws = _init_file2("Identity-By-State (IBS) analysis on genotypes:", gdsobj, sample_id, snp_id,
                     autosome_only, remove_monosnp, maf, missing_rate, num_thread, verbose, device)
    n = ws["n_samp"]
    o = [np.empty((n, n), np.int32) for _ in range(3)]
    _lib.check(_lib.lib().snpgpu_gnrIBSNum(ws["num_thread"], int(verbose), *[_lib._ptr(x) for x in o]))
    return dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], ibs0=o[0], ibs1=o[1], ibs2=o[2])


def snpgdsIBDKING(gdsobj, sample_id=None, snp_id=None, autosome_only=True, remove_monosnp=True,
                  maf=float("nan"), missing_rate=0.01, type="KING-robust", family_id=None,
                  num_thread=1, useMatrix=False, verbose=True, device=0):
    ws = _init_file2("IBD analysis (KING method of moment) on genotypes:", gdsobj, sample_id, snp_id,
                     autosome_only, remove_monosnp, maf, missing_rate, num_thread, verbose, device)
    if type not in ("KING-robust", "KING-homo"):
        raise ValueError("'arg' should be one of 'KING-robust', 'KING-homo'")   # match.arg
    n = ws["n_samp"]
    fam = None
    if family_id is not None:
        family_id = np.asarray(family_id)
        if n != len(family_id):
            raise ValueError("'length(family.id)' should be the number of samples.")
        if sample_id is not None:
            # family.id[match(sample.id, ws$sample.id)] (R/IBD.R:356-357), reproduced as it stands: the vector is
            # re-indexed by the position of each requested sample in the dataset-ordered working set
            ws_ids = np.asarray(ws["sample_id"])
            order = np.argsort(ws_ids, kind="stable")
            family_id = family_id[order[np.searchsorted(ws_ids[order], np.asarray(sample_id))]]
        # as.integer(as.factor(family.id)): every non-NA value is a level (negative integers too); "" and NA -> NA
        # (R/IBD.R:359-364).  -1 is only the ABI's code for NA after the factorisation.
        fam = np.full(n, -1, np.int32)
        if family_id.dtype.kind in "fc":
            good = ~np.isnan(family_id.astype(float))
        elif family_id.dtype.kind in "US":
            good = family_id != ""
        elif family_id.dtype.kind == "O":
            good = np.array([x is not None and x != "" for x in family_id])
        else:
            good = np.ones(n, bool)
        if good.any():
            _, codes = np.unique(family_id[good], return_inverse=True)
            fam[good] = codes.astype(np.int32) + 1
        _cat(verbose and type == "KING-robust", "# of families: %d, and within- and between-family "
             "relationship are estimated differently." % len(np.unique(fam[fam >= 0])))
    elif verbose and type == "KING-robust":
        print("No family is specified, and all individuals are treated as singletons.")
    a, b = _tri_or_full(n, useMatrix), _tri_or_full(n, useMatrix)
    rv = dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], afreq=None)
    if type == "KING-homo":
        _cat(verbose, "Relationship inference in a homogeneous population.")
        _lib.check(_lib.lib().snpgpu_gnrIBD_KING_Homo(ws["num_thread"], int(useMatrix), int(verbose),
                                                      _lib._ptr(a), _lib._ptr(b)))
        rv.update(k0=a, k1=b)
    else:
        _cat(verbose, "Relationship inference in the presence of population stratification.")
        _lib.check(_lib.lib().snpgpu_gnrIBD_KING_Robust(_lib._ptr(fam), ws["num_thread"], int(useMatrix),
                                                        int(verbose), _lib._ptr(a), _lib._ptr(b)))
        rv.update(IBS0=a, kinship=b)
    return rv


def snpgdsGRM(gdsobj, sample_id=None, snp_id=None, autosome_only=True, remove_monosnp=True,
              maf=float("nan"), missing_rate=0.01, method="GCTA", num_thread=1, useMatrix=False,
              out_fn=None, out_prec="double", out_compress="LZMA_RA", with_id=True, verbose=True, device=0):
    """snpgdsGRM (R/IBD.R:543-615).  out_fn: the reference writes a GDS file (FileFormat SNPRELATE_OUTPUT) through
    gdsfmt; gdsfmt is not available to this Python mirror, which stores THE SAME NODES (command, sample.id, snp.id,
    grm, avg_val) in a numpy archive under the given name -- readable by snpgdsMergeGRM here, NOT by the reference
    (and the reference's files are not readable here).  out_prec "double" / "single" selects the stored element
    type as in the reference; out_compress is accepted for signature compatibility and has no effect on the archive."""
    if out_prec not in ("double", "single"):
        raise ValueError("'arg' should be one of 'double', 'single'")     # match.arg
    all_methods = ("GCTA", "Eigenstrat", "EIGMIX", "Weighted", "Corr", "IndivBeta")
    if method not in all_methods:
        raise ValueError("'arg' should be one of " + ", ".join("'%s'" % m for m in all_methods))
    if method == "Weighted":          # R/IBD.R:552-556
        method = "EIGMIX"
    mtxt = {"Corr": "Scaled GCTA (correlation)", "EIGMIX": "EIGMIX / Weighted GCTA"}.get(method, method)
    ws = _init_file2("Genetic Relationship Matrix (GRM, %s):" % mtxt, gdsobj, sample_id, snp_id,
                     autosome_only, remove_monosnp, maf, missing_rate, num_thread, verbose, device)
    n = ws["n_samp"]
    # "Corr" always returns a full matrix (genPCA.cpp:1658); the output file always holds full rows (grm_save_to_gds)
    packed = bool(useMatrix) and method != "Corr" and out_fn is None
    out = _tri_or_full(n, packed)
    _lib.check(_lib.lib().snpgpu_gnrGRM(ws["num_thread"], method.encode(), int(packed), int(verbose),
                                        _lib._ptr(out)))
    avg = ctypes.c_double(0)
    if method == "IndivBeta":
        _lib.check(_lib.lib().snpgpu_gnrGRM_avg_val(ctypes.byref(avg)))
    if out_fn is not None:            # R/IBD.R:567-586,609-613: nodes of the SNPRELATE_OUTPUT file, nothing returned
        nodes = {"command": np.array(["snpgdsGRM", ":method = " + method]), "sample.id": ws["sample_id"],
                 "snp.id": ws["snp_id"], "grm": out if out_prec == "double" else out.astype(np.float32)}
        if method == "IndivBeta":
            nodes["avg_val"] = avg.value
        _gds.write_output(out_fn, nodes)
        return None
    if with_id:
        rv = dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], method=method, grm=out)
        if method == "IndivBeta":
            rv["avg_val"] = avg.value
        return rv
    return out


def snpgdsMergeGRM(filelist, out_fn=None, weight=None, verbose=True, device=0):
    """R/IBD.R:624-741 -> gnrGRMMerge (src/genPCA.cpp:1721-1853): combine the GRMs that snpgdsGRM(out_fn=) stored
    for disjoint SNP sets.  weight: None (by SNP count), a bool per file (False = subtract that SNP set) or numbers."""
    if isinstance(filelist, str) or len(filelist) == 0:
        raise ValueError("'filelist' should be a non-empty list of file names")
    if weight is not None and len(weight) != len(filelist):
        raise ValueError("length(weight) == length(filelist) is not TRUE")
    _cat(verbose, "GRM merging:")
    files = []
    for fn in filelist:
        f = _gds.read_output(fn)
        files.append(f)
        _cat(verbose, "    open '%s' (%s variants)" % (fn, format(len(f["snp.id"]), ",")))
    sampid = files[0]["sample.id"]
    dm = files[0]["grm"].shape
    if len(dm) != 2 or dm[0] != dm[1]:
        raise ValueError("'%s' has an invalid GRM matrix." % filelist[0])
    cmd = [str(x) for x in files[0]["command"]]
    if cmd[0] != "snpgdsGRM":
        raise ValueError("The GDS files should be created by snpgdsGRM()")
    for fn, f in zip(filelist, files):
        if [str(x) for x in f["command"]] != cmd:
            raise ValueError("'%s' has a different command." % fn)
        if f["grm"].shape != dm:
            raise ValueError("'%s' has a different GRM matrix." % fn)
    if weight is None or np.asarray(weight).dtype == np.bool_:
        num = np.array([float(len(f["snp.id"])) for f in files])
        if weight is not None:
            num[~np.asarray(weight, bool)] *= -1
        weight = num / num.sum()
    weight = np.ascontiguousarray(weight, np.float64)
    _cat(verbose, "Weight: " + ", ".join("%g" % w for w in weight))
    sid = np.array([], dtype=files[0]["snp.id"].dtype)
    for w, f in zip(weight, files):                             # R/IBD.R:704-712
        sid = np.concatenate([sid, f["snp.id"]]) if w >= 0 else sid[~np.isin(sid, f["snp.id"])]
    n = int(dm[0])
    beta = cmd[1] == ":method = IndivBeta"
    mats = [np.ascontiguousarray(f["grm"], np.float64) for f in files]
    ptrs = (ctypes.c_void_p * len(mats))(*[m.ctypes.data for m in mats])
    avg_in = np.ascontiguousarray([float(f["avg_val"]) for f in files], np.float64) if beta else None
    out = np.empty((n, n), np.float64)
    _lib.check(_lib.lib().snpgpu_gnrGRMMerge(len(mats), n, ptrs, cmd[1].encode(), _lib._ptr(avg_in) if beta else None,
                                             _lib._ptr(weight), _lib._ptr(out), int(device)))
    avg = ctypes.c_double(0)
    if beta:
        _lib.check(_lib.lib().snpgpu_gnrGRM_avg_val(ctypes.byref(avg)))
    if out_fn is not None:
        _cat(verbose, "Output: " + out_fn)
        nodes = {"command": np.array(cmd), "sample.id": sampid, "snp.id": sid, "grm": out}
        if beta:
            nodes["avg_val"] = avg.value
        _gds.write_output(out_fn, nodes)
        return None
    rv = dict(sample_id=sampid, snp_id=sid, grm=out)
    if beta:
        rv["avg_val"] = avg.value
    return rv


def snpgdsPCA(gdsobj, sample_id=None, snp_id=None, autosome_only=True, remove_monosnp=True,
              maf=float("nan"), missing_rate=0.01, algorithm="exact", eigen_cnt=None, num_thread=1,
              bayesian=False, need_genmat=False, genmat_only=False, eigen_method="DSPEVX",
              aux_dim=None, iter_num=10, aux_mat=None, verbose=True, device=0):
    """R/PCA.R:12-93.  algorithm = "exact" (covariance + eigen-decomposition) or "randomized" (Galinsky's
    fast PCA, CRandomPCA); aux_mat ([aux_dim][n_samp]) replaces R's rnorm(aux.dim * n.samp) when given."""
    if algorithm not in ("exact", "randomized"):
        raise ValueError("'arg' should be one of 'exact', 'randomized'")
    if eigen_cnt is None:
        eigen_cnt = 32 if algorithm == "exact" else 16          # R/PCA.R:15
    if eigen_method not in ("DSPEVX", "DSPEV"):
        raise ValueError("'arg' should be one of 'DSPEVX', 'DSPEV'")
    ws = _init_file2("Principal Component Analysis (PCA) on genotypes:", gdsobj, sample_id, snp_id,
                     autosome_only, remove_monosnp, maf, missing_rate, num_thread, verbose, device)
    n = ws["n_samp"]
    if algorithm == "randomized":
        if eigen_cnt <= 0:
            eigen_cnt = n
        if aux_dim is None:
            aux_dim = int(eigen_cnt) * 2                        # R/PCA.R:16
        if aux_mat is None:
            aux_mat = np.random.standard_normal((int(aux_dim), n))
        aux_mat = np.ascontiguousarray(aux_mat, np.float64)
        if aux_mat.shape != (int(aux_dim), n):
            raise ValueError("'aux.mat' should be aux.dim x n.samp")
        _cat(verbose, "    # of principal components: %d\n    starting from a random matrix [%d x %d]"
             % (eigen_cnt, aux_dim, n))
        d = np.empty(n, np.float64)
        ev = np.empty((int(eigen_cnt), n), np.float64)
        tr2 = ctypes.c_double(0)
        _lib.check(_lib.lib().snpgpu_gnrPCA_randomized(int(eigen_cnt), int(aux_dim), int(iter_num), _lib._ptr(aux_mat),
                                                       ws["num_thread"], int(verbose), _lib._ptr(d), _lib._ptr(ev),
                                                       ctypes.byref(tr2)))
        vp = 2 * d * d / tr2.value                              # R/PCA.R:82
        return dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], eigenval=(n - 1) * vp, eigenvect=ev.T,
                    varprop=vp, TraceXTX=tr2.value, Bayesian=False)
    if genmat_only:
        need_genmat = True
    if eigen_cnt <= 0:
        eigen_cnt = n
    eigen_cnt = min(int(eigen_cnt), n)
    _cat(verbose, "    # of principal components: %d" % eigen_cnt)
    genmat = np.empty((n, n), np.float64) if need_genmat else None
    tr, trv = ctypes.c_double(0), ctypes.c_double(0)
    eigval = eigvec = None
    if not genmat_only:
        eigval = np.empty(n, np.float64)
        eigvec = np.empty((eigen_cnt, n), np.float64)   # column-major n x k
    _lib.check(_lib.lib().snpgpu_gnrPCA(eigen_cnt, ws["num_thread"], int(bool(bayesian)), int(verbose),
                                        ctypes.byref(tr), _lib._ptr(genmat), _lib._ptr(eigval),
                                        _lib._ptr(eigvec), ctypes.byref(trv)))
    return dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], eigenval=eigval,
                eigenvect=None if eigvec is None else eigvec.T,
                varprop=None if eigval is None else eigval / trv.value,
                TraceXTX=tr.value, Bayesian=bool(bayesian), genmat=genmat)


def _init_file(gdsobj, sample_id=None, snp_id=None, device=0):
    """.InitFile, R/Internal.R:64-160: sample / SNP selection and gnrSetGenoSpace only (no filters)."""
    return _init_file2(None, gdsobj, sample_id, snp_id, autosome_only=False, remove_monosnp=False,
                       maf=float("nan"), missing_rate=float("nan"), num_thread=1, verbose=False, device=device)


def snpgdsPCACorr(pcaobj, gdsobj, snp_id=None, eig_which=None, num_thread=1, with_id=True, outgds=None,
                  verbose=True, device=0):
    """SNP correlations with the principal components (R/PCA.R:100-180 -> gnrPCACorr, src/genPCA.cpp:1455-1484).
    pcaobj: result of snpgdsPCA / snpgdsEIGMIX, or (sample_id, eigenvect [n][k]).  Returns snpcorr [k][n_snp]."""
    if outgds is not None:
        if not isinstance(outgds, str):
            raise TypeError("is.null(outgds) | is.character(outgds) is not TRUE")
        with_id = True
    if isinstance(pcaobj, dict):
        sampid, eigenvect = pcaobj["sample_id"], np.asarray(pcaobj["eigenvect"], np.float64)
    else:
        sampid, eigenvect = pcaobj
        eigenvect = np.asarray(eigenvect, np.float64)
    ws = _init_file(gdsobj, sampid, snp_id, device)
    if len(sampid) != eigenvect.shape[0]:
        raise ValueError("Internal error: the number of samples should be equal to the number of rows in 'eigenvect'.")
    if num_thread is None or num_thread <= 0:
        raise ValueError("num.thread > 0 is not TRUE")
    if eig_which is None:
        eig_which = np.arange(eigenvect.shape[1])
    else:
        eig_which = np.asarray(eig_which, dtype=np.int64) - 1          # R indices are 1-based
    _cat(verbose, "SNP Correlation:\n    # of samples: %d\n    # of SNPs: %d" % (ws["n_samp"], ws["n_snp"]))
    ev = np.ascontiguousarray(eigenvect[:, eig_which].T)               # [k][n] = n x k column-major
    out = np.empty((ws["n_snp"], ev.shape[0]), np.float64)             # k x n_snp column-major
    _lib.check(_lib.lib().snpgpu_gnrPCACorr(ev.shape[0], _lib._ptr(ev), int(num_thread), int(verbose), _lib._ptr(out)))
    if outgds is not None:
        # R/PCA.R:152-163: nodes sample.id, snp.id and "correlation" as packedreal16 (int16 steps of 1e-4, i.e. the
        # values read back are round(corr, 4), inst/unitTests/test_rel.R:148-152); nothing is returned
        _cat(verbose, "Creating '%s' ..." % outgds)
        q = np.where(np.isnan(out.T), np.nan, np.clip(np.round(out.T / 1e-4), -32767, 32767) * 1e-4)
        _gds.write_output(outgds, {"sample.id": np.asarray(sampid), "snp.id": ws["snp_id"], "correlation": q})
        return None
    if with_id:
        return dict(sample_id=np.asarray(sampid), snp_id=ws["snp_id"], snpcorr=out.T)
    return out.T


def snpgdsPCASNPLoading(pcaobj, gdsobj, num_thread=1, verbose=True, device=0):
    """SNP loadings (R/PCA.R:187-236 -> gnrPCASNPLoading, src/genPCA.cpp:1488-1531) of a snpgdsPCA result.
    Returns snploading [k][n_snp], avgfreq [n_snp] (mean genotype), scale [n_snp]."""
    if pcaobj.get("eigenval") is None or pcaobj.get("eigenvect") is None:
        raise ValueError("!is.null(pcaobj$eigenval), !is.null(pcaobj$eigenvect) are not all TRUE")
    ws = _init_file(gdsobj, pcaobj["sample_id"], pcaobj["snp_id"], device)
    ev = np.ascontiguousarray(np.asarray(pcaobj["eigenvect"], np.float64).T)    # [k][n]
    k = ev.shape[0]
    eigval = np.ascontiguousarray(np.asarray(pcaobj["eigenval"], np.float64)[:k])
    if "afreq" in pcaobj and "TraceXTX" not in pcaobj:
        # snpgdsEigMixClass, R/PCA.R:215-229 -> gnrEigMixSNPLoading, src/genEIGMIX.cpp:739-775
        if pcaobj.get("diagadj", False):
            raise ValueError("Please run `snpgdsEIGMIX(, diagadj=FALSE)` for projecting new samples.")
        af = np.ascontiguousarray(pcaobj["afreq"], np.float64)
        load = np.empty((ws["n_snp"], k), np.float64)
        _lib.check(_lib.lib().snpgpu_gnrEigMixSNPLoading(_lib._ptr(eigval), _lib._ptr(ev), k, _lib._ptr(af), int(num_thread),
                                                         int(verbose), _lib._ptr(load)))
        return dict(sample_id=np.asarray(pcaobj["sample_id"]), snp_id=np.asarray(pcaobj["snp_id"]),
                    eigenval=np.asarray(pcaobj["eigenval"]), snploading=load.T, afreq=af)
    _cat(verbose, "SNP Loading:\n    # of samples: %d\n    # of SNPs: %d\n    using the top %d eigenvectors"
         % (ws["n_samp"], ws["n_snp"], k))
    load = np.empty((ws["n_snp"], k), np.float64)
    af = np.empty(ws["n_snp"], np.float64)
    sc = np.empty(ws["n_snp"], np.float64)
    _lib.check(_lib.lib().snpgpu_gnrPCASNPLoading(_lib._ptr(eigval), _lib._ptr(ev), k, float(pcaobj["TraceXTX"]),
                                                  int(num_thread), int(bool(pcaobj.get("Bayesian", False))), int(verbose),
                                                  _lib._ptr(load), _lib._ptr(af), _lib._ptr(sc)))
    return dict(sample_id=np.asarray(pcaobj["sample_id"]), snp_id=np.asarray(pcaobj["snp_id"]),
                eigenval=np.asarray(pcaobj["eigenval"]), snploading=load.T, TraceXTX=pcaobj["TraceXTX"],
                Bayesian=bool(pcaobj.get("Bayesian", False)), avgfreq=af, scale=sc)


def snpgdsPCASampLoading(loadobj, gdsobj, sample_id=None, num_thread=1, verbose=True, device=0):
    """Project samples onto existing principal components (R/PCA.R:245-310 -> gnrPCASampLoading,
    src/genPCA.cpp:1535-1562).  Returns eigenvect [n_samp][k] (eigenval / varprop are NaN as in the reference)."""
    ws = _init_file(gdsobj, sample_id, loadobj["snp_id"], device)
    sl = np.asarray(loadobj["snploading"], np.float64)                  # [k][n_snp]
    k = sl.shape[0]
    if "avgfreq" not in loadobj:
        # snpgdsEigMixSNPLoadingClass, R/PCA.R:288-300 -> gnrEigMixSampLoading, src/genEIGMIX.cpp:777-803
        sqrt_eigval = np.sqrt(1 / np.asarray(loadobj["eigenval"], np.float64)[:k])
        sload = np.ascontiguousarray((sl * sqrt_eigval[:, None]).T)
        af = np.ascontiguousarray(loadobj["afreq"], np.float64)
        out = np.empty((k, ws["n_samp"]), np.float64)
        _lib.check(_lib.lib().snpgpu_gnrEigMixSampLoading(k, _lib._ptr(sload), _lib._ptr(af), int(num_thread), int(verbose),
                                                          _lib._ptr(out)))
        return dict(sample_id=ws["sample_id"], snp_id=np.asarray(loadobj["snp_id"]),
                    eigenval=np.full(ws["n_samp"], np.nan), eigenvect=out.T, afreq=af)
    _cat(verbose, "Sample Loading:\n    # of samples: %d\n    # of SNPs: %d\n    using the top %d eigenvectors"
         % (ws["n_samp"], ws["n_snp"], k))
    # prepare post-eigenvectors, R/PCA.R:281-285
    ss = (len(loadobj["sample_id"]) - 1) / loadobj["TraceXTX"]
    sqrt_eigval = np.sqrt(ss / np.asarray(loadobj["eigenval"], np.float64)[:k])
    sload = np.ascontiguousarray((sl * sqrt_eigval[:, None]).T)         # [n_snp][k] = k x n_snp column-major
    af = np.ascontiguousarray(loadobj["avgfreq"], np.float64)
    sc = np.ascontiguousarray(loadobj["scale"], np.float64)
    n = ws["n_samp"]
    out = np.empty((k, n), np.float64)                                  # n x k column-major
    _lib.check(_lib.lib().snpgpu_gnrPCASampLoading(k, _lib._ptr(sload), _lib._ptr(af), _lib._ptr(sc), int(num_thread),
                                                   int(verbose), _lib._ptr(out)))
    nan = np.full(n, np.nan)
    return dict(sample_id=ws["sample_id"], snp_id=np.asarray(loadobj["snp_id"]), eigenval=nan, eigenvect=out.T,
                varprop=nan.copy(), TraceXTX=loadobj["TraceXTX"], Bayesian=loadobj.get("Bayesian", False), genmat=None)


def snpgdsIBDMoM(gdsobj, sample_id=None, snp_id=None, autosome_only=True, remove_monosnp=True,
                 maf=float("nan"), missing_rate=0.01, allele_freq=None, kinship=False,
                 kinship_constraint=False, num_thread=1, useMatrix=False, verbose=True, device=0):
    """PLINK method of moments (R/IBD.R:22-68 -> gnrIBD_PLINK, src/genIBS.cpp:558-639)."""
    ws = _init_file2("IBD analysis (PLINK method of moment) on genotypes:", gdsobj, sample_id, snp_id,
                     autosome_only, remove_monosnp, maf, missing_rate, num_thread, verbose, device,
                     allele_freq=allele_freq)
    allele_freq = ws["allele_freq"]       # dataset order, filtered with gnrSelSNP_Base_Ex (R/Internal.R:355-452)
    n = ws["n_samp"]
    k0, k1 = _tri_or_full(n, useMatrix), _tri_or_full(n, useMatrix)
    af = np.empty(ws["n_snp"], np.float64)
    _lib.check(_lib.lib().snpgpu_gnrIBD_PLINK(ws["num_thread"], _lib._ptr(allele_freq), int(bool(kinship_constraint)),
                                              int(bool(useMatrix)), int(verbose), _lib._ptr(k0), _lib._ptr(k1),
                                              _lib._ptr(af)))
    af[af < 0] = np.nan
    ans = dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], afreq=af, k0=k0, k1=k1)
    if kinship:
        ans["kinship"] = 0.5 * (1 - k0 - k1) + 0.25 * k1
    return ans


def snpgdsIndivBeta(gdsobj, sample_id=None, snp_id=None, autosome_only=True, remove_monosnp=True,
                    maf=float("nan"), missing_rate=0.01, method="weighted", inbreeding=True, num_thread=1,
                    with_id=True, useMatrix=False, verbose=True, device=0):
    if method != "weighted":
        raise ValueError("'arg' should be one of 'weighted'")
    ws = _init_file2("Individual Inbreeding and Relatedness (beta estimator):", gdsobj, sample_id, snp_id,
                     autosome_only, remove_monosnp, maf, missing_rate, num_thread, verbose, device)
    out = _tri_or_full(ws["n_samp"], useMatrix)
    avg = ctypes.c_double(0)
    _lib.check(_lib.lib().snpgpu_gnrIBD_Beta(int(bool(inbreeding)), ws["num_thread"], int(bool(useMatrix)),
                                             int(verbose), _lib._ptr(out), ctypes.byref(avg)))
    if with_id:
        return dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], inbreeding=bool(inbreeding), beta=out,
                    avg_val=avg.value)
    return out


def snpgdsEIGMIX(gdsobj, sample_id=None, snp_id=None, autosome_only=True, remove_monosnp=True,
                 maf=float("nan"), missing_rate=0.01, num_thread=1, eigen_cnt=32, diagadj=True, ibdmat=False,
                 verbose=True, device=0):
    ws = _init_file2("Eigen-analysis on genotypes:", gdsobj, sample_id, snp_id, autosome_only, remove_monosnp,
                     maf, missing_rate, num_thread, verbose, device)
    n = ws["n_samp"]
    if eigen_cnt < 0:
        eigen_cnt = n
    k = min(int(eigen_cnt), n)
    ibd = np.empty((n, n), np.float64) if ibdmat else None
    eigval = np.empty(n, np.float64) if k > 0 else None
    eigvec = np.empty((k, n), np.float64) if k > 0 else None
    af = np.empty(ws["n_snp"], np.float64)
    _lib.check(_lib.lib().snpgpu_gnrEigMix(k, ws["num_thread"], int(bool(diagadj)), int(verbose), _lib._ptr(ibd),
                                           _lib._ptr(eigval), _lib._ptr(eigvec), _lib._ptr(af)))
    return dict(sample_id=ws["sample_id"], snp_id=ws["snp_id"], eigenval=eigval,
                eigenvect=None if eigvec is None else eigvec.T, afreq=af, ibd=ibd, diagadj=bool(diagadj))
