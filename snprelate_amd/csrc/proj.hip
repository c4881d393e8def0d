// C ABI of the PCA projections (include/snpgpu.h, "projector"): the bodies of gnrPCACorr, gnrPCASNPLoading and
// gnrPCASampLoading (src/genPCA.cpp:1455-1562) over one genotype block at a time; the block loop
// (CGenoReadBySNP) stays with the caller exactly as for the pairwise accumulators.
#include <cstring>

#include "snpgpu_internal.h"

using namespace snpgpu;

static inline int64_t up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

static void proj_free(snpgpu_proj *p)
{
    (void)hipSetDevice(p->device);
    DevBuf *all[] = {&p->raw, &p->packed, &p->sum, &p->num, &p->w2, &p->et, &p->eig_in, &p->out, &p->part, &p->cnt, &p->avg, &p->scale,
                     &p->sl, &p->af, &p->sc, &p->acc, &p->flag};
    for (DevBuf *b : all) b->release();
    if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

static int grow(DevBuf &b, size_t bytes)
{
    if (b.bytes >= bytes) return 0;
    b.release();
    return b.alloc(bytes);
}

// caller block -> packed rows + statistics (+ sample-major words when `words`)
static int stage_block(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem, bool words)
{
    if (!geno) {       // reuse the block staged by the previous call (same n_snp): no copy, no repack
        if (n_snp != p->staged_snps) { set_error("snpgpu_proj: no staged block of this size to reuse"); return 1; }
        if (words && !p->staged_words) {
            const int64_t n_pad = up(n_snp, 64);
            if (launch_transpose2(p->stream, (const uint8_t *)p->packed.p, p->RB, n_snp, 0, p->ncols_pad, (int)(n_pad / 16),
                                  (uint32_t *)p->w2.p, nullptr, nullptr, true))
                return 1;
            p->staged_words = true;
        }
        return 0;
    }
    if (n_snp <= 0 || n_snp > p->Bmax) { set_error("snpgpu_proj: invalid block (empty or larger than max_block_snps)"); return 1; }
    p->staged_snps = 0;
    if (format != SNPGPU_GENO_U8 && format != SNPGPU_GENO_PACKED2) { set_error("snpgpu_proj: invalid format"); return 1; }
    const size_t in_bytes = (size_t)n_snp * (size_t)(format == SNPGPU_GENO_U8 ? p->N : (p->N + 3) / 4);
    const void *src = geno;
    if (mem != SNPGPU_DEVICE) {
        if (grow(p->raw, in_bytes)) return 1;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(p->raw.p, geno, in_bytes, hipMemcpyHostToDevice, p->stream));
        src = p->raw.p;
    }
    SNPGPU_HIP_CHECK(hipMemsetAsync(p->flag.p, 0, 8, p->stream));
    if (launch_repack_stats(p->stream, src, format, n_snp, p->N, (uint8_t *)p->packed.p, p->RB, (int32_t *)p->sum.p,
                            (int32_t *)p->num.p, (unsigned long long *)p->flag.p))
        return 1;
    if (words) {
        const int64_t n_pad = up(n_snp, 64);
        if (launch_transpose2(p->stream, (const uint8_t *)p->packed.p, p->RB, n_snp, 0, p->ncols_pad, (int)(n_pad / 16),
                              (uint32_t *)p->w2.p, nullptr, nullptr, true))
            return 1;
    }
    p->staged_snps = n_snp;
    p->staged_words = words;
    return 0;
}

static int copy_out(snpgpu_proj *p, void *user, const void *dev, size_t bytes, int mem)
{
    if (!user) return 0;
    SNPGPU_HIP_CHECK(hipMemcpyAsync(user, dev, bytes, mem == SNPGPU_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                                    p->stream));
    return 0;
}

extern "C" {

int snpgpu_proj_create(int64_t n_samp, int n_eig, const snpgpu_opts *opts, snpgpu_proj **out)
{
    if (!out) { set_error("snpgpu_proj_create: out is NULL"); return 1; }
    *out = nullptr;
    if (n_samp <= 0 || n_samp > 0x7fffffffLL) { set_error("snpgpu_proj_create: invalid number of samples"); return 1; }
    if (n_eig <= 0 || n_eig > 4096) { set_error("snpgpu_proj_create: invalid number of eigenvectors"); return 1; }
    snpgpu_opts o{};
    if (opts) o = *opts;
    int ndev = 0;
    SNPGPU_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) { set_error("snpgpu_proj_create: no HIP device (the GPU path has no CPU fallback)"); return 1; }
    if (o.device < 0 || o.device >= ndev) { set_error("snpgpu_proj_create: invalid device ordinal"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(o.device));
    snpgpu_proj *p = new snpgpu_proj();
    p->device = o.device; p->N = n_samp; p->k = n_eig; p->kp = (int)up(n_eig, 16);
    p->RB = up(n_samp, 256) / 4;
    p->ncols_pad = up(n_samp, 256);
    p->n_pad = p->ncols_pad;
    p->Bmax = up(o.max_block_snps > 0 ? o.max_block_snps : 16384, 64);
    if (o.stream) p->stream = (hipStream_t)o.stream;
    else if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) == hipSuccess) p->own_stream = true;
    else { set_error("snpgpu_proj_create: hipStreamCreate failed"); delete p; return 1; }
    int rc = 0;
    rc |= p->packed.alloc((size_t)p->Bmax * (size_t)p->RB);
    rc |= p->sum.alloc(4 * (size_t)p->Bmax);
    rc |= p->num.alloc(4 * (size_t)p->Bmax);
    rc |= p->w2.alloc(4 * (size_t)(p->Bmax / 16) * (size_t)p->ncols_pad);
    rc |= p->et.alloc(8 * (size_t)p->n_pad * (size_t)p->kp);
    rc |= p->out.alloc(8 * (size_t)p->Bmax * (size_t)p->k);
    rc |= p->part.alloc(8 * 3 * (size_t)p->Bmax * (size_t)p->k);
    rc |= p->cnt.alloc(4 * 3 * (size_t)p->Bmax);
    rc |= p->avg.alloc(8 * (size_t)p->Bmax);
    rc |= p->scale.alloc(8 * (size_t)p->Bmax);
    rc |= p->sl.alloc(8 * (size_t)p->Bmax * (size_t)p->kp);
    rc |= p->af.alloc(8 * (size_t)p->Bmax);
    rc |= p->sc.alloc(8 * (size_t)p->Bmax);
    rc |= p->acc.alloc(8 * (size_t)p->N * (size_t)p->k);
    rc |= p->flag.alloc(64);
    if (!rc && hipMemsetAsync(p->acc.p, 0, p->acc.bytes, p->stream) != hipSuccess) rc = 1;
    if (!rc && hipStreamSynchronize(p->stream) != hipSuccess) rc = 1;
    if (rc) { proj_free(p); set_error("snpgpu_proj_create: device allocation failed"); return 1; }
    *out = p;
    return 0;
}

int snpgpu_proj_destroy(snpgpu_proj *p)
{
    if (!p) return 0;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    proj_free(p);
    return 0;
}

int snpgpu_proj_sync(snpgpu_proj *p)
{
    if (!p) { set_error("snpgpu_proj_sync: NULL projector"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(p->device));
    SNPGPU_HIP_CHECK(hipStreamSynchronize(p->stream));
    return 0;
}

int snpgpu_proj_set_eigvec(snpgpu_proj *p, const double *eigvec, int mem)
{
    if (!p || !eigvec) { set_error("snpgpu_proj_set_eigvec: NULL argument"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(p->device));
    const size_t bytes = 8 * (size_t)p->N * (size_t)p->k;
    const double *src = eigvec;
    if (mem != SNPGPU_DEVICE) {
        if (grow(p->eig_in, bytes)) return 1;
        SNPGPU_HIP_CHECK(hipMemcpyAsync(p->eig_in.p, eigvec, bytes, hipMemcpyHostToDevice, p->stream));
        src = (const double *)p->eig_in.p;
    }
    if (launch_proj_transpose(p->stream, src, p->N, p->k, (double *)p->et.p, p->n_pad, p->kp)) return 1;
    if (mem != SNPGPU_DEVICE) SNPGPU_HIP_CHECK(hipStreamSynchronize(p->stream));
    p->have_eig = true;
    return 0;
}

static int snp_side(snpgpu_proj *p, int corr, const void *geno, int64_t n_snp, int format, int mem, int bayesian,
                    double *out, double *afreq, double *scale, int out_mem, const char *fn,
                    const double *ext_avg = nullptr, const double *ext_scale = nullptr, int ext_mem = SNPGPU_HOST)
{
    if (!p) { set_error(std::string(fn) + ": NULL projector"); return 1; }
    if (!p->have_eig) { set_error(std::string(fn) + ": call snpgpu_proj_set_eigvec first"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(p->device));
    if (stage_block(p, geno, n_snp, format, mem, true)) return 1;
    double *d_out = (out_mem == SNPGPU_DEVICE && out) ? out : (double *)p->out.p;
    double *d_avg = (out_mem == SNPGPU_DEVICE && afreq) ? afreq : (double *)p->avg.p;
    double *d_sc = (out_mem == SNPGPU_DEVICE && scale) ? scale : (double *)p->scale.p;
    const double *x_avg = nullptr, *x_sc = nullptr;
    if (ext_avg) {                                  // caller-supplied centring / scaling
        if (ext_mem == SNPGPU_DEVICE) { x_avg = ext_avg; x_sc = ext_scale; }
        else {
            SNPGPU_HIP_CHECK(hipMemcpyAsync(p->af.p, ext_avg, 8 * (size_t)n_snp, hipMemcpyHostToDevice, p->stream));
            SNPGPU_HIP_CHECK(hipMemcpyAsync(p->sc.p, ext_scale, 8 * (size_t)n_snp, hipMemcpyHostToDevice, p->stream));
            x_avg = (const double *)p->af.p; x_sc = (const double *)p->sc.p;
        }
    }
    if (launch_proj_snp(p->stream, corr, (const uint32_t *)p->w2.p, p->ncols_pad, p->N, n_snp, (const double *)p->et.p,
                        p->kp, p->k, (const int32_t *)p->sum.p, (const int32_t *)p->num.p, bayesian, d_out, (double *)p->part.p,
                        (int *)p->cnt.p, d_avg, d_sc, x_avg, x_sc))
        return 1;
    if (out_mem != SNPGPU_DEVICE) {
        if (copy_out(p, out, d_out, 8 * (size_t)n_snp * (size_t)p->k, out_mem)) return 1;
        if (!corr) {
            if (copy_out(p, afreq, d_avg, 8 * (size_t)n_snp, out_mem)) return 1;
            if (copy_out(p, scale, d_sc, 8 * (size_t)n_snp, out_mem)) return 1;
        }
        SNPGPU_HIP_CHECK(hipStreamSynchronize(p->stream));
    } else if (mem != SNPGPU_DEVICE) {
        SNPGPU_HIP_CHECK(hipStreamSynchronize(p->stream));     // the caller may reuse its host block
    }
    return 0;
}

int snpgpu_proj_snp_corr(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem, double *out, int out_mem)
{
    return snp_side(p, 1, geno, n_snp, format, mem, 0, out, nullptr, nullptr, out_mem, "snpgpu_proj_snp_corr");
}

int snpgpu_proj_snp_loading(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem, int bayesian,
                            double *loading, double *afreq, double *scale, int out_mem)
{
    return snp_side(p, 0, geno, n_snp, format, mem, bayesian, loading, afreq, scale, out_mem, "snpgpu_proj_snp_loading");
}

int snpgpu_proj_snp_loading_ext(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem, const double *avg,
                                const double *scale, int in_mem, double *loading, int out_mem)
{
    if (!avg || !scale) { set_error("snpgpu_proj_snp_loading_ext: NULL argument"); return 1; }
    return snp_side(p, 0, geno, n_snp, format, mem, 0, loading, nullptr, nullptr, out_mem, "snpgpu_proj_snp_loading_ext", avg,
                    scale, in_mem);
}

int snpgpu_proj_samp_loading_feed(snpgpu_proj *p, const void *geno, int64_t n_snp, int format, int mem,
                                  const double *sload, const double *afreq, const double *scale, int in_mem)
{
    if (!p || !sload || !afreq || !scale) { set_error("snpgpu_proj_samp_loading_feed: NULL argument"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(p->device));
    if (stage_block(p, geno, n_snp, format, mem, false)) return 1;
    const hipMemcpyKind kind = (in_mem == SNPGPU_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (p->kp != p->k) SNPGPU_HIP_CHECK(hipMemsetAsync(p->sl.p, 0, 8 * (size_t)n_snp * (size_t)p->kp, p->stream));
    SNPGPU_HIP_CHECK(hipMemcpy2DAsync(p->sl.p, 8 * (size_t)p->kp, sload, 8 * (size_t)p->k, 8 * (size_t)p->k, (size_t)n_snp,
                                      kind, p->stream));
    SNPGPU_HIP_CHECK(hipMemcpyAsync(p->af.p, afreq, 8 * (size_t)n_snp, kind, p->stream));
    SNPGPU_HIP_CHECK(hipMemcpyAsync(p->sc.p, scale, 8 * (size_t)n_snp, kind, p->stream));
    if (launch_proj_samp(p->stream, (const uint8_t *)p->packed.p, p->RB, p->N, n_snp, (const double *)p->sl.p, p->kp, p->k,
                         (const double *)p->af.p, (const double *)p->sc.p, (double *)p->acc.p))
        return 1;
    if (mem != SNPGPU_DEVICE || in_mem != SNPGPU_DEVICE) SNPGPU_HIP_CHECK(hipStreamSynchronize(p->stream));
    return 0;
}

int snpgpu_proj_samp_loading_reset(snpgpu_proj *p)
{
    if (!p) { set_error("snpgpu_proj_samp_loading_reset: NULL projector"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(p->device));
    SNPGPU_HIP_CHECK(hipMemsetAsync(p->acc.p, 0, p->acc.bytes, p->stream));
    return 0;
}

int snpgpu_proj_samp_loading(snpgpu_proj *p, double *out, int out_mem)
{
    if (!p || !out) { set_error("snpgpu_proj_samp_loading: NULL argument"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(p->device));
    if (copy_out(p, out, p->acc.p, 8 * (size_t)p->N * (size_t)p->k, out_mem)) return 1;
    SNPGPU_HIP_CHECK(hipStreamSynchronize(p->stream));
    return 0;
}

}  // extern "C"
