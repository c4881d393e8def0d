// One host process, several GPUs: the multi-device accumulator behind the C ABI (include/snpgpu.h, "snpgpu_multi").
//
// north_star: "the N x N output triangle is row-block partitioned across the 8 GPUs of one node with a final gather over
// xGMI".  An R session is ONE process, so the R shim cannot use the one-process-per-GPU drivers of bench.py; this object
// gives it the same plan from inside the library:
//   * the packed triangle is cut into equal-area row panels (plan_* below: the C++ twin of snprelate_amd/dist.py, which
//     restates Array_SplitJobs, src/dGenGWAS.cpp:2202-2216, across devices), `panels_per_device` per device (the last
//     equal-area panel is a square holding a triangle: several panels per device even out the memory), optionally only
//     the panels of one PASS of several (KING-robust's 20 B per pair at N = 500 000 do not fit a node at once);
//   * every feed block crosses PCIe ONCE, to the first device, and is forwarded to the others over xGMI
//     (hipMemcpyPeerAsync on per-device copy streams, double-buffered, overlapping the kernels of the previous block);
//     there is no collective on the data path;
//   * finalisers gather the panels' packed slabs -- contiguous ranges of the packed triangle -- into the caller's host
//     buffer, or into device memory of the first device through peer copies;
//   * the top-k eigen solver runs its tall-skinny algebra on the first device and the O(N^2) product on every device:
//     the vector block is broadcast, the partial products are reduced -- with RCCL (ncclBroadcast / ncclReduce on a
//     communicator of all devices, librccl loaded at run time) when the devices are distinct, with peer copies + an add
//     kernel otherwise (tests put several "devices" on one GPU).
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "eigen.h"

using namespace snpgpu;

namespace {

inline int64_t tri_offset(int64_t n, int64_t i) { return i * n - i * (i - 1) / 2; }
inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// row boundaries of `parts` panels of equal TIME, interior boundaries multiples of 256: the twin of dist.panel_rows (same
// arithmetic, same order of operations).  cost(panel [r0, r1)) = pairs + alpha * (n - r0): the pre-pass transposes the panel's
// columns r0 .. n of every block; alpha = 512 pairs per column (measured ~500 for GRM, ~360 for the counters; SNPGPU_PLAN_ALPHA)
double plan_alpha()
{
    if (const char *e = getenv("SNPGPU_PLAN_ALPHA")) {
        char *end = nullptr;
        const double v = strtod(e, &end);
        if (end != e && v >= 0.0) return v;
    }
    return 512.0;
}

std::vector<int64_t> plan_rows(int64_t n, int parts)
{
    const double nf = (double)n, alpha = std::min(plan_alpha(), nf / (4.0 * parts));    // (capped as in dist.panel_rows)
    auto tri = [&](double b) { return b * nf - b * (b - 1.0) / 2.0; };
    auto ends = [&](double T) {
        std::vector<double> b{0.0};
        for (int p = 0; p < parts; p++) {
            const double r0 = b.back(), budget = T - alpha * (nf - r0);
            double x = r0;
            if (budget > 0.0) {
                const double disc = (2.0 * nf + 1.0) * (2.0 * nf + 1.0) - 8.0 * (tri(r0) + budget);
                x = disc <= 0.0 ? nf : ((2.0 * nf + 1.0) - std::sqrt(disc)) / 2.0;
                x = std::min(std::max(x, r0), nf);
            }
            b.push_back(x);
        }
        return b;
    };
    double lo = 0.0, hi = tri(nf) + alpha * nf;
    for (int it = 0; it < 100; it++) {
        const double mid = 0.5 * (lo + hi);
        if (ends(mid).back() >= nf) hi = mid; else lo = mid;
    }
    const std::vector<double> cont = ends(hi);
    std::vector<int64_t> b{0};
    for (int r = 1; r < parts; r++) {
        int64_t v = (int64_t)std::nearbyint(cont[(size_t)r] / PANEL_ALIGN) * PANEL_ALIGN;
        v = std::min<int64_t>(std::max<int64_t>(v, b.back()), n / PANEL_ALIGN * PANEL_ALIGN);
        b.push_back(v);
    }
    b.push_back(n);
    return b;
}

int64_t plan_storage(int64_t n, int64_t r0, int64_t r1) { return r1 > r0 ? round_up(r1 - r0, PANEL_ALIGN) * round_up(n - r0, PANEL_ALIGN) : 0; }

// owned[pass][device] = sorted panel indices: largest storage first into the least loaded slot (dist.pass_plan)
std::vector<std::vector<std::vector<int>>> plan_owners(int64_t n, const std::vector<int64_t> &bounds, int n_dev, int ppd, int passes)
{
    const int P = (int)bounds.size() - 1;
    std::vector<int64_t> size((size_t)P);
    for (int p = 0; p < P; p++) size[(size_t)p] = plan_storage(n, bounds[(size_t)p], bounds[(size_t)p + 1]);
    std::vector<int> order((size_t)P);
    for (int p = 0; p < P; p++) order[(size_t)p] = p;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return size[(size_t)a] > size[(size_t)b]; });
    std::vector<std::vector<std::vector<int>>> owned((size_t)passes, std::vector<std::vector<int>>((size_t)n_dev));
    std::vector<std::vector<int64_t>> load((size_t)passes, std::vector<int64_t>((size_t)n_dev, 0));
    for (int p : order) {
        int bq = -1, bd = -1;
        for (int q = 0; q < passes; q++)
            for (int d = 0; d < n_dev; d++) {
                if ((int)owned[(size_t)q][(size_t)d].size() >= ppd) continue;
                if (bq < 0 || load[(size_t)q][(size_t)d] < load[(size_t)bq][(size_t)bd] ||
                    (load[(size_t)q][(size_t)d] == load[(size_t)bq][(size_t)bd] &&
                     owned[(size_t)q][(size_t)d].size() < owned[(size_t)bq][(size_t)bd].size())) {
                    bq = q; bd = d;
                }
            }
        owned[(size_t)bq][(size_t)bd].push_back(p);
        load[(size_t)bq][(size_t)bd] += size[(size_t)p];
    }
    for (auto &q : owned)
        for (auto &d : q) std::sort(d.begin(), d.end());
    return owned;
}

// panels_per_device = -1: the fewest panels per device that fit.  One equal-area panel per device leaves the LAST device a square
// holding a triangle (twice the mean storage: 233 GiB of fp64 at N = 500 000 on 8 devices), several panels per device even that
// out -- but every panel is a context with its own feed-block scratch (2-bit rows, sample-major words, tables: ~ max_block_snps x
// (N / 4 + 0.95 x the panel's columns) bytes for a GRM, 39 GB at N = 500 000 and 65 536-SNP blocks), so more panels are not free.
// Per device: sum over its panels of elements x bytes per pair + scratch, + two forwarding buffers, against the device's free memory
// less 4 GiB (a device listed k times gets 1 / k of it).  First with the eigen solver's fp32 copy of the panel (GRM / PCA / EIGMIX
// kinds: + 4 bytes per pair), then without.
int64_t panel_scratch_bytes(int kind, int64_t n, int64_t bmax, int64_t r0)
{
    const int64_t np = round_up(n - r0, PANEL_ALIGN), rb = round_up(n, 256) / 4, bp = round_up(bmax, 1024);
    int64_t b = bmax * rb;                                                           // packed
    const bool pc = kind != SNPGPU_PCA_COV && kind != SNPGPU_EIGMIX, mm = kind == SNPGPU_KING_HOMO || kind == SNPGPU_GRM_GCTA ||
                    kind == SNPGPU_PCA_COV || kind == SNPGPU_EIGMIX;
    if (pc) b += 4 * (bmax / 16 + 32) * np;                                          // w2
    if (kind == SNPGPU_GRM_GCTA) b += bmax * np / 8;                                 // mm256
    if (mm) b += 4 * (bp / 8 + 96) * np + 8 * (5 * bp / 512 + 16) * np;              // wt, tcorr
    if (kind == SNPGPU_EIGMIX) b += 4 * (bp / 8 + 96) * np;                          // wt12
    if (kind == SNPGPU_KING_HOMO) b += 16 * (bp / 256 + 16) * np + 16 * np + 2 * 64 * (bp + 2048) + 16 * (bp + 2048);   // homo_tc, homo_msum, homo_lut x 2, homo_wts
    if (mm && kind != SNPGPU_KING_HOMO)                                              // per-SNP tables of the single-product / exact-row kernels
        b += (64 + 64 + 32 + 32 + 8 + 16 + 8 * UV_QMAX) * (bp + 2048) + 16 * (bmax + 2048) + 16 * np + 8 * np;     // (uvlut, uvpace, ...)
    return b;
}

int auto_panels_per_device(int kind, int64_t n, int64_t bmax, const int32_t *devices, int nd, int passes, std::string *why, int at_least = 1)
{
    const double per_pair[] = {12, 20, 24, 12, 8, 32, 12};      // IBS, KING-robust, KING-homo, GCTA (8 + 4), PCA, EIGMIX (2 x 8 + ...), beta
    const double bpe = (kind >= SNPGPU_IBS && kind <= SNPGPU_INDIV_BETA) ? per_pair[kind - SNPGPU_IBS] : 8;
    const bool eig = kind == SNPGPU_GRM_GCTA || kind == SNPGPU_PCA_COV || kind == SNPGPU_EIGMIX;
    std::vector<double> budget((size_t)nd, 0.0);
    for (int d = 0; d < nd; d++) {
        size_t fr = 0, tot = 0;
        if (hipSetDevice(devices[d]) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) { *why = "cannot query device memory"; return 0; }
        int listed = 0;
        for (int e = 0; e < nd; e++) listed += devices[e] == devices[d];
        budget[(size_t)d] = ((double)fr - 4.0 * 1073741824.0) / listed - 2.0 * (double)bmax * (double)((n + 3) / 4);
    }
    const int tries[] = {1, 2, 3, 4, 6, 8, 12, 16};
    for (int with_copy = eig ? 1 : 0; with_copy >= 0; with_copy--)
        for (int ppd : tries) {
            if (ppd < at_least) continue;
            const std::vector<int64_t> b = plan_rows(n, nd * ppd * passes);
            const auto owned = plan_owners(n, b, nd, ppd, passes);
            bool fits = true;
            for (int q = 0; q < passes && fits; q++)
                for (int d = 0; d < nd && fits; d++) {
                    double need = 0;
                    for (int p : owned[(size_t)q][(size_t)d])
                        if (b[(size_t)p + 1] > b[(size_t)p])
                            need += (double)plan_storage(n, b[(size_t)p], b[(size_t)p + 1]) * (bpe + (with_copy ? 4.0 : 0.0)) +
                                    (double)panel_scratch_bytes(kind, n, bmax, b[(size_t)p]);
                    fits = need <= budget[(size_t)d];
                }
            if (fits) return ppd;
        }
    *why = "the accumulators of " + std::to_string(n) + " samples do not fit " + std::to_string(nd) + " device(s) in " + std::to_string(passes) +
           " pass(es) with up to 16 panels per device: lower max_block_snps (the per-panel scratch grows with it) or raise n_passes";
    return 0;
}

// byte-per-genotype rows [n_snp][N] -> 2-bit rows [n_snp][(N + 3) / 4] (the GDS bit2 layout: sample 4 b + k at bits 2 k; values
// above 2 = missing = 3, as CGenoReadBySNP clamps them): what the first device forwards to its peers is a quarter of what the
// kept byte-inflating reader delivered
__global__ __launch_bounds__(256) void u8_to_packed2_kernel(const uint8_t *__restrict__ src, int64_t N, int64_t rb,
                                                            uint8_t *__restrict__ dst)
{
    const uint8_t *__restrict__ row = src + (int64_t)blockIdx.x * N;
    uint8_t *__restrict__ out = dst + (int64_t)blockIdx.x * rb;
    for (int64_t b = (int64_t)blockIdx.y * 256 + threadIdx.x; b < rb; b += (int64_t)gridDim.y * 256) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t s = 4 * b + k;
            const uint32_t g = s < N ? (uint32_t)row[s] : 3u;
            v |= (g > 2u ? 3u : g) << (2 * k);
        }
        out[b] = (uint8_t)v;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(double *__restrict__ y, const double *__restrict__ x, double f, size_t n)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) y[e] = f * x[e];
}

__global__ __launch_bounds__(256) void add_kernel(double *__restrict__ y, const double *__restrict__ x, size_t n)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) y[e] += x[e];
}

// ---- RCCL, loaded at run time (the library has no link-time dependency on it) ------------------------------------------
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, void *, hipStream_t) = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Broadcast = (decltype(Broadcast))dlsym(lib, "ncclBroadcast");
        Reduce = (decltype(Reduce))dlsym(lib, "ncclReduce");
        return CommInitAll && CommDestroy && GroupStart && GroupEnd && Broadcast && Reduce;
    }
};
Rccl g_rccl;
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;       // ncclFloat64, ncclSum (rccl.h)

struct Dev {
    int device = 0;
    hipStream_t copy = nullptr;                   // forwards the feed blocks; also the eigen operator's stream on this device
    DevBuf blk[2];                                // double-buffered copy of the current feed block
    DevBuf raw[2];                                // first device only: a host block of byte genotypes before it is packed
    hipEvent_t ready[2] = {nullptr, nullptr};     // blk[s] holds its block
    std::vector<int> panels;                      // indices into snpgpu_multi::ctx
    DevBuf q, y;                                  // eigen operator: this device's copy of the vector block, its partial product
};

}  // namespace

struct snpgpu_multi {
    int kind = 0;
    int64_t N = 0, Bmax = 0;
    std::vector<Dev> dev;
    std::vector<snpgpu_ctx *> ctx;                // one per resident panel
    std::vector<int> ctx_dev;                     // index into `dev`
    std::vector<int> panel_index;                 // index in the whole plan
    std::vector<std::vector<hipEvent_t>> used;    // [ctx][slot]: the context's pre-pass has read blk[slot]
    std::vector<int64_t> bounds;                  // row boundaries of ALL panels of the plan
    int turn = 0;
    const void *host_src[2] = {nullptr, nullptr};
    std::vector<void *> comms;                    // RCCL communicators (one per device) or empty
    bool frozen_checked = false;
    int ppd = 1;                                  // panels per device of the plan (the resolved value when the caller asked for -1)
    int peer_pairs = 0, peer_enabled = 0;         // ordered pairs of distinct devices / of those with peer access switched on
    int st_comm = -1, st_feed = -1, st_gather = -1;   // snpgpu_multi_comm_selftest: -1 not run, 0 failed, 1 passed
};

namespace {

void multi_free(snpgpu_multi *m)
{
    for (snpgpu_ctx *c : m->ctx)
        if (c) snpgpu_destroy(c);
    for (size_t i = 0; i < m->used.size(); i++) {
        (void)hipSetDevice(m->dev[(size_t)m->ctx_dev[i]].device);
        for (hipEvent_t e : m->used[i])
            if (e) (void)hipEventDestroy(e);
    }
    for (size_t d = 0; d < m->dev.size(); d++) {
        Dev &D = m->dev[d];
        (void)hipSetDevice(D.device);
        if (d < m->comms.size() && m->comms[d] && g_rccl.CommDestroy) g_rccl.CommDestroy(m->comms[d]);
        if (D.copy) (void)hipStreamSynchronize(D.copy);
        D.blk[0].release(); D.blk[1].release(); D.raw[0].release(); D.raw[1].release(); D.q.release(); D.y.release();
        for (int s = 0; s < 2; s++)
            if (D.ready[s]) (void)hipEventDestroy(D.ready[s]);
        if (D.copy) (void)hipStreamDestroy(D.copy);
    }
    delete m;
}

// the vector block goes to every device, the partial products come back summed: y = scale * C q over all panels
class MultiOperator : public EigOperator {
public:
    MultiOperator(snpgpu_multi *m, double scale) : m_(m), scale_(scale) {}
    int64_t n() const override { return m_->N; }
    int device() const override { return m_->dev[0].device; }
    int apply(const double *Q, int b, double *Y, bool fp32_products) override
    {
        const size_t count = (size_t)b * (size_t)m_->N, bytes = sizeof(double) * count;
        const size_t nd = m_->dev.size();
        Dev &D0 = m_->dev[0];
        const bool rccl = !m_->comms.empty();
        for (size_t d = 1; d < nd; d++) {
            Dev &D = m_->dev[d];
            SNPGPU_HIP_CHECK(hipSetDevice(D.device));
            if (D.q.bytes < bytes) { D.q.release(); D.y.release(); if (D.q.alloc(bytes) || D.y.alloc(bytes)) return 1; }
        }
        if (rccl) {
            if (g_rccl.GroupStart()) return fail("ncclGroupStart");
            for (size_t d = 0; d < nd; d++) {
                Dev &D = m_->dev[d];
                if (g_rccl.Broadcast(Q, d == 0 ? (void *)Q : D.q.p, count, NCCL_DOUBLE, 0, m_->comms[d], D.copy)) return fail("ncclBroadcast");
            }
            if (g_rccl.GroupEnd()) return fail("ncclGroupEnd");
        } else {
            for (size_t d = 1; d < nd; d++) {
                Dev &D = m_->dev[d];
                SNPGPU_HIP_CHECK(hipSetDevice(D.device));
                SNPGPU_HIP_CHECK(hipMemcpyPeerAsync(D.q.p, D.device, Q, D0.device, bytes, D.copy));
            }
        }
        for (size_t d = 0; d < nd; d++) {          // clear the partial products, then let the panels' streams go
            Dev &D = m_->dev[d];
            SNPGPU_HIP_CHECK(hipSetDevice(D.device));
            SNPGPU_HIP_CHECK(hipMemsetAsync(d == 0 ? (void *)Y : D.y.p, 0, bytes, D.copy));
        }
        for (size_t d = 0; d < nd; d++) {
            SNPGPU_HIP_CHECK(hipSetDevice(m_->dev[d].device));
            SNPGPU_HIP_CHECK(hipStreamSynchronize(m_->dev[d].copy));
        }
        for (size_t i = 0; i < m_->ctx.size(); i++) {
            const size_t d = (size_t)m_->ctx_dev[i];
            if (ctx_panel_matmul_enqueue(m_->ctx[i], scale_, d == 0 ? Q : (const double *)m_->dev[d].q.p, b,
                                         d == 0 ? Y : (double *)m_->dev[d].y.p, fp32_products))
                return 1;
        }
        for (size_t i = 0; i < m_->ctx.size(); i++) {
            SNPGPU_HIP_CHECK(hipSetDevice(m_->ctx[i]->device));
            SNPGPU_HIP_CHECK(hipStreamSynchronize(m_->ctx[i]->stream));
        }
        if (rccl) {
            if (g_rccl.GroupStart()) return fail("ncclGroupStart");
            for (size_t d = 0; d < nd; d++) {
                Dev &D = m_->dev[d];
                const void *send = d == 0 ? (const void *)Y : (const void *)D.y.p;
                if (g_rccl.Reduce(send, d == 0 ? (void *)Y : D.y.p, count, NCCL_DOUBLE, NCCL_SUM, 0, m_->comms[d], D.copy)) return fail("ncclReduce");
            }
            if (g_rccl.GroupEnd()) return fail("ncclGroupEnd");
            for (size_t d = 0; d < nd; d++) {
                SNPGPU_HIP_CHECK(hipSetDevice(m_->dev[d].device));
                SNPGPU_HIP_CHECK(hipStreamSynchronize(m_->dev[d].copy));
            }
        } else if (nd > 1) {
            SNPGPU_HIP_CHECK(hipSetDevice(D0.device));
            if (D0.q.bytes < bytes) { D0.q.release(); if (D0.q.alloc(bytes)) return 1; }     // staging of a peer's partial product
            for (size_t d = 1; d < nd; d++) {
                SNPGPU_HIP_CHECK(hipMemcpyPeerAsync(D0.q.p, D0.device, m_->dev[d].y.p, m_->dev[d].device, bytes, D0.copy));
                hipLaunchKernelGGL(add_kernel, dim3(2048), dim3(256), 0, D0.copy, Y, (const double *)D0.q.p, count);
            }
            SNPGPU_HIP_CHECK(hipStreamSynchronize(D0.copy));
        }
        SNPGPU_HIP_CHECK(hipSetDevice(D0.device));
        return 0;
    }

private:
    int fail(const char *what) { set_error(std::string("snpgpu_multi: ") + what + " failed"); return 1; }
    snpgpu_multi *m_;
    double scale_;
};

// gather the packed slabs of `n_out` results (element size `esz`) into the caller's buffers: `fin` finalises one panel
// into device buffers on the panel's device
template <class Fin>
int gather_slabs(snpgpu_multi *m, int n_out, size_t esz, void *const *out, int mem, Fin fin)
{
    if (mem != SNPGPU_HOST && mem != SNPGPU_DEVICE) { set_error("snpgpu_multi: results go to host memory or to device memory of the first device"); return 1; }
    // Round 6: every device finalises and ships its own panels CONCURRENTLY -- one host thread per device entry (the finalisers
    // synchronise their context's stream, so one thread would serialise the devices as the round-5 loop did), the slab straight
    // into its range of the destination through an asynchronous peer / device-to-host copy on the device's copy stream.  Errors
    // are thread-local in this library: each thread hands its message back.  One temporary slab per device at a time, as before.
    const int dev0 = m->dev[0].device;
    const size_t nd = m->dev.size();
    std::vector<std::string> err(nd);
    std::vector<int> trc(nd, 0);
    auto work = [&](size_t d) {
        Dev &D = m->dev[d];
        if (hipSetDevice(D.device) != hipSuccess) { trc[d] = 1; err[d] = "hipSetDevice failed"; return; }
        for (int i : D.panels) {
            snpgpu_ctx *c = m->ctx[(size_t)i];
            const size_t elems = (size_t)snpgpu_slab_size(c);
            const size_t off = (size_t)tri_offset(m->N, c->row0);
            std::vector<DevBuf> tmp((size_t)n_out);
            std::vector<void *> ptr((size_t)n_out);
            const bool direct = (mem == SNPGPU_DEVICE && c->device == dev0);
            int rc = 0;
            for (int k = 0; k < n_out && !rc; k++) {
                if (direct) ptr[(size_t)k] = (char *)out[k] + off * esz;
                else { rc = tmp[(size_t)k].alloc(elems * esz); ptr[(size_t)k] = tmp[(size_t)k].p; }
            }
            if (!rc) rc = fin(c, ptr.data());
            if (rc) err[d] = snpgpu_last_error();
            for (int k = 0; k < n_out && !rc && !direct; k++) {
                hipError_t e = (mem == SNPGPU_HOST) ? hipMemcpyAsync((char *)out[k] + off * esz, ptr[(size_t)k], elems * esz, hipMemcpyDeviceToHost, D.copy)
                                                    : hipMemcpyPeerAsync((char *)out[k] + off * esz, dev0, ptr[(size_t)k], c->device, elems * esz, D.copy);
                if (e != hipSuccess) { err[d] = std::string("snpgpu_multi: gather copy failed: ") + hipGetErrorString(e); rc = 1; }
            }
            if (!direct) {
                const hipError_t e = hipStreamSynchronize(D.copy);
                if (e != hipSuccess && !rc) { err[d] = std::string("snpgpu_multi: gather copy failed: ") + hipGetErrorString(e); rc = 1; }
            }
            for (DevBuf &t : tmp) t.release();
            if (rc) { trc[d] = 1; return; }
        }
    };
    if (nd == 1 || getenv("SNPGPU_MULTI_GATHER_SERIAL")) {
        for (size_t d = 0; d < nd; d++) work(d);
    } else {
        std::vector<std::thread> th;
        for (size_t d = 0; d < nd; d++) th.emplace_back(work, d);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(dev0);
    for (size_t d = 0; d < nd; d++)
        if (trc[d]) { set_error(err[d].empty() ? "snpgpu_multi: gather failed" : err[d]); return 1; }
    return 0;
}

int need(snpgpu_multi *m, int kind_a, int kind_b, const char *fn)
{
    if (!m) { set_error(std::string(fn) + ": NULL object"); return 1; }
    if (m->kind != kind_a && m->kind != kind_b) { set_error(std::string(fn) + ": wrong kind"); return 1; }
    return 0;
}

}  // namespace

extern "C" {

static int multi_create_with(int kind, int64_t n_samp, const snpgpu_opts *opts, const snpgpu_multi_opts *mo, int ppd, snpgpu_multi **out);

int snpgpu_multi_create(int kind, int64_t n_samp, const snpgpu_opts *opts, const snpgpu_multi_opts *mo, snpgpu_multi **out)
{
    if (!out) { set_error("snpgpu_multi_create: out is NULL"); return 1; }
    *out = nullptr;
    if (!mo || !mo->devices || mo->n_devices <= 0) { set_error("snpgpu_multi_create: no device list"); return 1; }
    const int passes = mo->n_passes > 0 ? mo->n_passes : 1;
    if (mo->pass < 0 || mo->pass >= passes) { set_error("snpgpu_multi_create: invalid pass"); return 1; }
    if (n_samp <= 0) { set_error("snpgpu_multi_create: invalid number of samples"); return 1; }
    if (mo->panels_per_device >= 0) return multi_create_with(kind, n_samp, opts, mo, mo->panels_per_device > 0 ? mo->panels_per_device : 1, out);
    // automatic: the fewest panels per device whose accumulators AND per-panel scratch fit by the estimate (auto_panels_per_device);
    // the estimate is not the allocator -- when a context of that plan still fails with "out of memory" the next larger count is
    // tried instead of failing the job
    const int64_t bmax = round_up(opts && opts->max_block_snps > 0 ? opts->max_block_snps : 32768, 64);
    for (int at_least = 1;;) {
        std::string why;
        const int ppd = auto_panels_per_device(kind, n_samp, bmax, mo->devices, mo->n_devices, passes, &why, at_least);
        if (ppd <= 0) { set_error("snpgpu_multi_create: " + why); return 1; }
        if (!multi_create_with(kind, n_samp, opts, mo, ppd, out)) return 0;
        const std::string err = snpgpu_last_error();
        if (err.find("out of memory") == std::string::npos || ppd >= 16) return 1;
        fprintf(stderr, "snpgpu_multi_create: %d panel(s) per device did not fit after all (%s); trying more, smaller panels\n", ppd, err.c_str());
        at_least = ppd + 1;
    }
}

static int multi_create_with(int kind, int64_t n_samp, const snpgpu_opts *opts, const snpgpu_multi_opts *mo, int ppd, snpgpu_multi **out)
{
    const int passes = mo->n_passes > 0 ? mo->n_passes : 1;
    snpgpu_opts o{};
    if (opts) o = *opts;
    if (o.stream) { set_error("snpgpu_multi_create: a caller stream cannot serve several devices"); return 1; }
    std::unique_ptr<snpgpu_multi, void (*)(snpgpu_multi *)> m(new snpgpu_multi(), multi_free);
    m->kind = kind; m->N = n_samp; m->ppd = ppd;
    m->Bmax = round_up(o.max_block_snps > 0 ? o.max_block_snps : 32768, 64);
    const int nd = mo->n_devices;
    m->bounds = plan_rows(n_samp, nd * ppd * passes);
    const auto owned = plan_owners(n_samp, m->bounds, nd, ppd, passes);
    m->dev.resize((size_t)nd);
    for (int d = 0; d < nd; d++) {
        Dev &D = m->dev[(size_t)d];
        D.device = mo->devices[d];
        SNPGPU_HIP_CHECK(hipSetDevice(D.device));
        SNPGPU_HIP_CHECK(hipStreamCreateWithFlags(&D.copy, hipStreamNonBlocking));
        for (int s = 0; s < 2; s++) SNPGPU_HIP_CHECK(hipEventCreateWithFlags(&D.ready[s], hipEventDisableTiming));
        // peer access for the forwarding copies, the gathers and the eigen exchanges.  Checked (round 6): without it the runtime
        // stages every peer copy through host memory -- correct, several times slower -- and the deployment should know
        for (int e = 0; e < nd; e++) {
            if (mo->devices[e] == D.device) continue;
            bool seen = false;                     // a device listed more than once: count each ordered pair of distinct devices once
            for (int f = 0; f < e; f++) seen = seen || mo->devices[f] == mo->devices[e];
            for (int f = 0; f < d; f++) seen = seen || mo->devices[f] == D.device;
            int can = 0;
            hipError_t pe = hipDeviceCanAccessPeer(&can, D.device, mo->devices[e]);
            if (pe == hipSuccess && can) {
                pe = hipDeviceEnablePeerAccess(mo->devices[e], 0);
                if (pe == hipErrorPeerAccessAlreadyEnabled) pe = hipSuccess;
            } else if (pe == hipSuccess) pe = hipErrorPeerAccessUnsupported;
            (void)hipGetLastError();
            if (!seen) {
                m->peer_pairs++;
                if (pe == hipSuccess) m->peer_enabled++;
                else
                    fprintf(stderr, "snpgpu_multi_create: no peer access from device %d to device %d (%s): copies between them are staged through "
                                    "host memory by the runtime (snpgpu_multi_get_status reports peer_pairs_enabled)\n", D.device, mo->devices[e],
                            hipGetErrorString(pe));
            }
        }
        for (int p : owned[(size_t)mo->pass][(size_t)d]) {
            const int64_t r0 = m->bounds[(size_t)p], r1 = m->bounds[(size_t)p + 1];
            if (r1 <= r0) continue;
            snpgpu_opts po = o;
            po.device = D.device;
            po.row_begin = (r0 == 0 && r1 == n_samp) ? 0 : r0;
            po.row_end = (r0 == 0 && r1 == n_samp) ? 0 : r1;
            snpgpu_ctx *c = nullptr;
            if (snpgpu_create(kind, n_samp, &po, &c)) return 1;
            D.panels.push_back((int)m->ctx.size());
            m->ctx.push_back(c);
            m->ctx_dev.push_back(d);
            m->panel_index.push_back(p);
            std::vector<hipEvent_t> ev(2, nullptr);
            for (int s = 0; s < 2; s++) SNPGPU_HIP_CHECK(hipEventCreateWithFlags(&ev[(size_t)s], hipEventDisableTiming));
            m->used.push_back(ev);
        }
    }
    // (a pass may own no panel at all when n is small against the plan -- 256-row boundaries --: feeds and gathers are then no-ops)
    // RCCL communicator for the eigen solver's broadcast / reduce when the devices are distinct (SNPGPU_MULTI_COMM=peer:
    // peer copies instead; =rccl: insist)
    const char *want = getenv("SNPGPU_MULTI_COMM");
    bool distinct = true;
    for (int a = 0; a < nd; a++)
        for (int b = a + 1; b < nd; b++)
            if (mo->devices[a] == mo->devices[b]) distinct = false;
    const bool insist = want && std::string(want) == "rccl";
    if (distinct && (nd > 1 || insist) && !(want && std::string(want) == "peer")) {
        if (g_rccl.load()) {
            m->comms.assign((size_t)nd, nullptr);
            const int rc = g_rccl.CommInitAll(m->comms.data(), nd, mo->devices);
            if (rc != 0) {
                m->comms.clear();
                if (insist) { set_error("snpgpu_multi_create: ncclCommInitAll failed on " + std::to_string(nd) + " devices (ncclResult " + std::to_string(rc) + ")"); return 1; }
                // never silent: the eigen solver's exchanges then use peer copies + an add kernel on the first device
                fprintf(stderr, "snpgpu_multi_create: ncclCommInitAll failed on %d devices (ncclResult %d); falling back to peer copies "
                                "(SNPGPU_MULTI_COMM=rccl makes this an error, snpgpu_multi_comm_selftest checks the path in use)\n", nd, rc);
            }
        } else if (insist) { set_error("snpgpu_multi_create: librccl could not be loaded"); return 1; }
        else fprintf(stderr, "snpgpu_multi_create: librccl could not be loaded; the eigen solver's exchanges use peer copies\n");
    } else if (insist) { set_error("snpgpu_multi_create: RCCL needs distinct devices"); return 1; }
    *out = m.release();
    return 0;
}

int snpgpu_multi_destroy(snpgpu_multi *m)
{
    if (m) multi_free(m);
    return 0;
}

int snpgpu_multi_info(const snpgpu_multi *m, int *n_panels, int *uses_rccl)
{
    if (!m) { set_error("snpgpu_multi_info: NULL object"); return 1; }
    if (n_panels) *n_panels = (int)m->ctx.size();
    if (uses_rccl) *uses_rccl = m->comms.empty() ? 0 : 1;
    return 0;
}

int snpgpu_multi_get_status(const snpgpu_multi *m, snpgpu_multi_status *out)
{
    if (!m || !out) { set_error("snpgpu_multi_get_status: NULL argument"); return 1; }
    memset(out, 0, sizeof(*out));
    out->n_devices = (int32_t)m->dev.size();
    std::vector<int> ds;
    for (const Dev &D : m->dev) ds.push_back(D.device);
    std::sort(ds.begin(), ds.end());
    out->n_distinct_devices = (int32_t)(std::unique(ds.begin(), ds.end()) - ds.begin());
    out->n_panels = (int32_t)m->ctx.size();
    out->panels_per_device = m->ppd;
    out->uses_rccl = m->comms.empty() ? 0 : 1;
    out->peer_pairs = m->peer_pairs;
    out->peer_pairs_enabled = m->peer_enabled;
    out->selftest_comm = m->st_comm;
    out->selftest_feed = m->st_feed;
    out->selftest_gather = m->st_gather;
    return 0;
}

namespace {

__global__ __launch_bounds__(256) void pattern_kernel(uint32_t *__restrict__ p, size_t n, uint32_t salt)
{
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
        p[e] = ((uint32_t)e * 2654435761u) ^ salt;
}

// sum of (word ^ expected): zero iff every word is the expected one
__global__ __launch_bounds__(256) void pattern_check_kernel(const uint32_t *__restrict__ p, size_t n, uint32_t salt, unsigned long long *__restrict__ bad)
{
    unsigned long long b = 0;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256)
        b += (p[e] != (((uint32_t)e * 2654435761u) ^ salt));
    if (b) atomicAdd(bad, b);
}

// The two data paths of the object that the eigen exchange test does not touch, with a known pattern
// and a per-device count of wrong words:
//   feed   -- a 2-bit block on the first device forwarded to every other device exactly as snpgpu_multi_feed does it
//             (hipMemcpyPeerAsync on the receiving device's copy stream, ordered by the first device's `ready` event), verified ON
//             each receiving device;
//   gather -- a slab from every device written into its range of one buffer on the first device exactly as the gathers do it
//             (peer copy issued by the sending device's thread on its copy stream), verified on the first device.
int path_selftest(snpgpu_multi *m, std::string *why)
{
    const size_t nd = m->dev.size(), words = 1u << 18, bytes = words * 4;       // 1 MiB per device
    std::vector<DevBuf> buf(nd), flag(nd);
    DevBuf all;
    hipEvent_t ev = nullptr;
    auto cleanup = [&]() {
        for (size_t d = 0; d < nd; d++) { (void)hipSetDevice(m->dev[d].device); buf[d].release(); flag[d].release(); }
        (void)hipSetDevice(m->dev[0].device);
        all.release();
        if (ev) (void)hipEventDestroy(ev);
    };
    auto fail = [&](const std::string &w) { cleanup(); *why = w; return 1; };
    m->st_feed = m->st_gather = 0;
    for (size_t d = 0; d < nd; d++)
        if (hipSetDevice(m->dev[d].device) != hipSuccess || buf[d].alloc(bytes) || flag[d].alloc(8) ||
            hipMemsetAsync(flag[d].p, 0, 8, m->dev[d].copy) != hipSuccess || hipMemsetAsync(buf[d].p, 0xA5, bytes, m->dev[d].copy) != hipSuccess)
            return fail("allocation failed");
    Dev &D0 = m->dev[0];
    (void)hipSetDevice(D0.device);
    if (all.alloc(bytes * nd) || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail("allocation failed");
    // ---- feed: pattern on the first device, forwarded star-wise
    hipLaunchKernelGGL(pattern_kernel, dim3(64), dim3(256), 0, D0.copy, (uint32_t *)buf[0].p, words, 0x2b17u);
    if (hipEventRecord(ev, D0.copy) != hipSuccess) return fail("event record failed");
    for (size_t d = 1; d < nd; d++) {
        Dev &D = m->dev[d];
        (void)hipSetDevice(D.device);
        if (hipStreamWaitEvent(D.copy, ev, 0) != hipSuccess ||
            hipMemcpyPeerAsync(buf[d].p, D.device, buf[0].p, D0.device, bytes, D.copy) != hipSuccess)
            return fail("feed path: peer copy to device " + std::to_string(D.device) + " could not be enqueued");
    }
    std::vector<unsigned long long> bad(nd, 0);
    for (size_t d = 0; d < nd; d++) {
        Dev &D = m->dev[d];
        (void)hipSetDevice(D.device);
        hipLaunchKernelGGL(pattern_check_kernel, dim3(64), dim3(256), 0, D.copy, (const uint32_t *)buf[d].p, words, 0x2b17u, (unsigned long long *)flag[d].p);
        if (hipMemcpyAsync(&bad[d], flag[d].p, 8, hipMemcpyDeviceToHost, D.copy) != hipSuccess || hipStreamSynchronize(D.copy) != hipSuccess)
            return fail("feed path: device " + std::to_string(D.device) + " did not complete");
        if (bad[d]) return fail("feed path: " + std::to_string(bad[d]) + " of " + std::to_string(words) + " words wrong on device " + std::to_string(D.device) +
                                " (entry " + std::to_string(d) + " of the device list)");
    }
    m->st_feed = 1;
    // ---- gather: every device's slab (its own salt) into its range on the first device, one host thread per device as in gather_slabs
    std::vector<int> trc(nd, 0);
    std::vector<std::thread> th;
    for (size_t d = 0; d < nd; d++)
        th.emplace_back([&, d]() {
            Dev &D = m->dev[d];
            if (hipSetDevice(D.device) != hipSuccess) { trc[d] = 1; return; }
            hipLaunchKernelGGL(pattern_kernel, dim3(64), dim3(256), 0, D.copy, (uint32_t *)buf[d].p, words, 0x9e00u + (uint32_t)d);
            hipError_t e = hipMemcpyPeerAsync((char *)all.p + bytes * d, D0.device, buf[d].p, D.device, bytes, D.copy);
            if (e == hipSuccess) e = hipStreamSynchronize(D.copy);
            trc[d] = e != hipSuccess;
        });
    for (auto &t : th) t.join();
    for (size_t d = 0; d < nd; d++)
        if (trc[d]) return fail("gather path: copy from device " + std::to_string(m->dev[d].device) + " failed");
    (void)hipSetDevice(D0.device);
    if (hipMemsetAsync(flag[0].p, 0, 8, D0.copy) != hipSuccess) return fail("memset failed");
    for (size_t d = 0; d < nd; d++) {
        unsigned long long b = 0;
        hipLaunchKernelGGL(pattern_check_kernel, dim3(64), dim3(256), 0, D0.copy, (const uint32_t *)((char *)all.p + bytes * d), words, 0x9e00u + (uint32_t)d,
                           (unsigned long long *)flag[0].p);
        if (hipMemcpyAsync(&b, flag[0].p, 8, hipMemcpyDeviceToHost, D0.copy) != hipSuccess || hipStreamSynchronize(D0.copy) != hipSuccess)
            return fail("gather path: check did not complete");
        if (b) return fail("gather path: " + std::to_string(b) + " words wrong in the slab of device " + std::to_string(m->dev[d].device));
    }
    m->st_gather = 1;
    cleanup();
    return 0;
}

}  // namespace

// One broadcast + one sum-reduction of a known pattern over the object's devices, through the path the eigen solver will use
// (RCCL communicator or peer copies): device d multiplies the broadcast pattern by d + 1, the first device must receive
// pattern * nd (nd + 1) / 2.  Fails loudly -- the first collective of a deployment should not be the 500 000-sample job's.
int snpgpu_multi_comm_selftest(snpgpu_multi *m, int *uses_rccl)
{
    if (!m) { set_error("snpgpu_multi_comm_selftest: NULL object"); return 1; }
    if (uses_rccl) *uses_rccl = m->comms.empty() ? 0 : 1;
    const size_t nd = m->dev.size(), count = 4096, bytes = count * sizeof(double);
    const bool rccl = !m->comms.empty();
    m->st_comm = 0;
    std::vector<double> h(count);
    for (size_t e = 0; e < count; e++) h[e] = 1.0 + (double)(e % 97) * 0.25;
    std::vector<DevBuf> q(nd), y(nd);
    auto cleanup = [&]() { for (size_t d = 0; d < nd; d++) { (void)hipSetDevice(m->dev[d].device); q[d].release(); y[d].release(); } };
    auto fail = [&](const std::string &what) { cleanup(); set_error("snpgpu_multi_comm_selftest: " + what); return 1; };
    for (size_t d = 0; d < nd; d++) {
        if (hipSetDevice(m->dev[d].device) != hipSuccess || q[d].alloc(bytes) || y[d].alloc(bytes)) return fail("allocation failed");
        if (hipMemsetAsync(q[d].p, 0, bytes, m->dev[d].copy) != hipSuccess) return fail("memset failed");
    }
    (void)hipSetDevice(m->dev[0].device);
    if (hipMemcpyAsync(q[0].p, h.data(), bytes, hipMemcpyHostToDevice, m->dev[0].copy) != hipSuccess ||
        hipStreamSynchronize(m->dev[0].copy) != hipSuccess) return fail("upload failed");
    if (rccl) {
        if (g_rccl.GroupStart()) return fail("ncclGroupStart failed");
        for (size_t d = 0; d < nd; d++)
            if (g_rccl.Broadcast(q[0].p, q[d].p, count, NCCL_DOUBLE, 0, m->comms[d], m->dev[d].copy)) return fail("ncclBroadcast failed");
        if (g_rccl.GroupEnd()) return fail("ncclGroupEnd failed");
    } else {
        for (size_t d = 1; d < nd; d++) {
            (void)hipSetDevice(m->dev[d].device);
            if (hipMemcpyPeerAsync(q[d].p, m->dev[d].device, q[0].p, m->dev[0].device, bytes, m->dev[d].copy) != hipSuccess) return fail("peer copy failed");
        }
    }
    for (size_t d = 0; d < nd; d++) {
        (void)hipSetDevice(m->dev[d].device);
        hipLaunchKernelGGL(scale_kernel, dim3(16), dim3(256), 0, m->dev[d].copy, (double *)y[d].p, (const double *)q[d].p, (double)(d + 1), count);
        if (hipStreamSynchronize(m->dev[d].copy) != hipSuccess) return fail("device " + std::to_string(m->dev[d].device) + " did not complete");
    }
    if (rccl) {
        if (g_rccl.GroupStart()) return fail("ncclGroupStart failed");
        for (size_t d = 0; d < nd; d++)
            if (g_rccl.Reduce(y[d].p, y[d].p, count, NCCL_DOUBLE, NCCL_SUM, 0, m->comms[d], m->dev[d].copy)) return fail("ncclReduce failed");
        if (g_rccl.GroupEnd()) return fail("ncclGroupEnd failed");
        for (size_t d = 0; d < nd; d++) { (void)hipSetDevice(m->dev[d].device); if (hipStreamSynchronize(m->dev[d].copy) != hipSuccess) return fail("reduce did not complete"); }
    } else {
        (void)hipSetDevice(m->dev[0].device);
        for (size_t d = 1; d < nd; d++) {
            if (hipMemcpyPeerAsync(q[0].p, m->dev[0].device, y[d].p, m->dev[d].device, bytes, m->dev[0].copy) != hipSuccess) return fail("peer copy failed");
            hipLaunchKernelGGL(add_kernel, dim3(16), dim3(256), 0, m->dev[0].copy, (double *)y[0].p, (const double *)q[0].p, count);
        }
        if (hipStreamSynchronize(m->dev[0].copy) != hipSuccess) return fail("reduce did not complete");
    }
    std::vector<double> got(count);
    (void)hipSetDevice(m->dev[0].device);
    if (hipMemcpy(got.data(), y[0].p, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail("download failed");
    const double f = 0.5 * (double)nd * (double)(nd + 1);
    for (size_t e = 0; e < count; e++)
        if (got[e] != h[e] * f) return fail("wrong sum at element " + std::to_string(e) + " (" + (rccl ? "RCCL" : "peer copies") + ", " + std::to_string(nd) + " devices)");
    cleanup();
    m->st_comm = 1;
    // the feed-forward star and the gather path (round 6)
    std::string why;
    if (path_selftest(m, &why)) { set_error("snpgpu_multi_comm_selftest: " + why); return 1; }
    return 0;
}

int snpgpu_multi_panel(const snpgpu_multi *m, int i, snpgpu_ctx **ctx, int64_t *row_begin, int64_t *row_end, int *device)
{
    if (!m || i < 0 || i >= (int)m->ctx.size()) { set_error("snpgpu_multi_panel: invalid panel"); return 1; }
    if (ctx) *ctx = m->ctx[(size_t)i];
    if (row_begin) *row_begin = m->ctx[(size_t)i]->row0;
    if (row_end) *row_end = m->ctx[(size_t)i]->row1;
    if (device) *device = m->ctx[(size_t)i]->device;
    return 0;
}

int snpgpu_multi_feed(snpgpu_multi *m, const void *geno, int64_t n_snp, int format, int mem)
{
    if (!m) { set_error("snpgpu_multi_feed: NULL object"); return 1; }
    if (n_snp == 0) return 0;
    if (!geno || n_snp < 0 || n_snp > m->Bmax) { set_error("snpgpu_multi_feed: invalid block (larger than max_block_snps?)"); return 1; }
    if (format != SNPGPU_GENO_U8 && format != SNPGPU_GENO_PACKED2) { set_error("snpgpu_multi_feed: invalid format"); return 1; }
    if (m->ctx.empty()) return 0;
    // With more than one device a block of byte genotypes is packed to 2-bit rows on the first device before anything is
    // forwarded: the star below then carries N / 4 bytes per SNP and peer instead of N (configs[2] on 8 GPUs, 32 768-SNP blocks
    // every ~42 ms: 20 GB/s per link instead of 78 -- more than PCIe or one xGMI link carries), and the panels are fed
    // SNPGPU_GENO_PACKED2.  One device: the block is consumed in the format it came in.
    const bool pack = (format == SNPGPU_GENO_U8 && m->dev.size() > 1);
    const int fmt = pack ? SNPGPU_GENO_PACKED2 : format;
    const size_t row_in = (size_t)(format == SNPGPU_GENO_U8 ? m->N : (m->N + 3) / 4), row = (size_t)(fmt == SNPGPU_GENO_U8 ? m->N : (m->N + 3) / 4);
    const size_t bytes_in = (size_t)n_snp * row_in, cap_in = (size_t)m->Bmax * row_in;
    const size_t bytes = (size_t)n_snp * row, cap = (size_t)m->Bmax * row;
    const int s = m->turn;
    m->turn ^= 1;
    Dev &D0 = m->dev[0];
    // (1) the block on the first device: the caller's device memory as it is, host memory through one PCIe copy
    const void *src0 = geno;
    SNPGPU_HIP_CHECK(hipSetDevice(D0.device));
    if (mem != SNPGPU_DEVICE || pack) {
        if (D0.blk[s].bytes < cap) { SNPGPU_HIP_CHECK(hipStreamSynchronize(D0.copy)); D0.blk[s].release(); if (D0.blk[s].alloc(cap)) return 1; }
        // blk[s] was last read by the first device's own contexts and by the peers' forwarding copies, two blocks ago
        for (size_t i = 0; i < m->ctx.size(); i++)
            if (m->ctx_dev[i] == 0) SNPGPU_HIP_CHECK(hipStreamWaitEvent(D0.copy, m->used[i][(size_t)s], 0));
        for (size_t d = 1; d < m->dev.size(); d++) SNPGPU_HIP_CHECK(hipStreamWaitEvent(D0.copy, m->dev[d].ready[s], 0));
        const void *in = geno;
        if (mem != SNPGPU_DEVICE) {
            DevBuf &dst = pack ? D0.raw[s] : D0.blk[s];       // (raw[s] is read by the packing kernel on this same stream only)
            if (pack && dst.bytes < cap_in) { SNPGPU_HIP_CHECK(hipStreamSynchronize(D0.copy)); dst.release(); if (dst.alloc(cap_in)) return 1; }
            SNPGPU_HIP_CHECK(hipMemcpyAsync(dst.p, geno, bytes_in, hipMemcpyHostToDevice, D0.copy));
            in = dst.p;
            m->host_src[s] = geno;
        }
        if (pack) {
            const int64_t rb = (int64_t)row;
            hipLaunchKernelGGL(u8_to_packed2_kernel, dim3((unsigned)n_snp, (unsigned)std::min<int64_t>((rb + 255) / 256, 64)), dim3(256), 0, D0.copy,
                               (const uint8_t *)in, m->N, rb, (uint8_t *)D0.blk[s].p);
            SNPGPU_HIP_CHECK(hipGetLastError());
        }
        src0 = D0.blk[s].p;
    }
    SNPGPU_HIP_CHECK(hipEventRecord(D0.ready[s], D0.copy));
    // (2) forwarded to every other device over xGMI
    for (size_t d = 1; d < m->dev.size(); d++) {
        Dev &D = m->dev[d];
        SNPGPU_HIP_CHECK(hipSetDevice(D.device));
        if (D.blk[s].bytes < cap) { SNPGPU_HIP_CHECK(hipStreamSynchronize(D.copy)); D.blk[s].release(); if (D.blk[s].alloc(cap)) return 1; }
        SNPGPU_HIP_CHECK(hipStreamWaitEvent(D.copy, D0.ready[s], 0));
        for (int i : D.panels) SNPGPU_HIP_CHECK(hipStreamWaitEvent(D.copy, m->used[(size_t)i][(size_t)s], 0));
        SNPGPU_HIP_CHECK(hipMemcpyPeerAsync(D.blk[s].p, D.device, src0, D0.device, bytes, D.copy));
        SNPGPU_HIP_CHECK(hipEventRecord(D.ready[s], D.copy));
    }
    // (3) every resident panel consumes its device's copy
    for (size_t i = 0; i < m->ctx.size(); i++) {
        Dev &D = m->dev[(size_t)m->ctx_dev[i]];
        snpgpu_ctx *c = m->ctx[i];
        SNPGPU_HIP_CHECK(hipSetDevice(D.device));
        SNPGPU_HIP_CHECK(hipStreamWaitEvent(c->stream, D.ready[s], 0));
        const void *src = (m->ctx_dev[i] == 0) ? src0 : D.blk[s].p;
        if (snpgpu_feed(c, src, n_snp, fmt, SNPGPU_DEVICE)) return 1;
        SNPGPU_HIP_CHECK(hipEventRecord(m->used[i][(size_t)s], c->stream));
    }
    if (mem == SNPGPU_HOST) {                     // pageable memory: the caller may reuse its buffer on return
        SNPGPU_HIP_CHECK(hipSetDevice(D0.device));
        SNPGPU_HIP_CHECK(hipStreamSynchronize(D0.copy));
    }
    return 0;
}

int snpgpu_multi_host_wait(snpgpu_multi *m, const void *host_buf)
{
    if (!m) { set_error("snpgpu_multi_host_wait: NULL object"); return 1; }
    SNPGPU_HIP_CHECK(hipSetDevice(m->dev[0].device));
    for (int s = 0; s < 2; s++)
        if (m->host_src[s] == host_buf) SNPGPU_HIP_CHECK(hipEventSynchronize(m->dev[0].ready[s]));
    return 0;
}

int snpgpu_multi_sync(snpgpu_multi *m)
{
    if (!m) return 0;
    for (Dev &D : m->dev) {
        SNPGPU_HIP_CHECK(hipSetDevice(D.device));
        SNPGPU_HIP_CHECK(hipStreamSynchronize(D.copy));
    }
    for (snpgpu_ctx *c : m->ctx)
        if (snpgpu_sync(c)) return 1;
    return 0;
}

int snpgpu_multi_counts(snpgpu_multi *m, int64_t *n_snp_total, int64_t *n_locus)
{
    if (!m) { set_error("snpgpu_multi_counts: NULL object"); return 1; }
    if (m->ctx.empty()) { if (n_snp_total) *n_snp_total = 0; if (n_locus) *n_locus = 0; return 0; }
    return snpgpu_counts(m->ctx[0], n_snp_total, n_locus);
}

// ---- gathers (packed triangle; rows of panels that are not resident in this pass are left untouched) -------------------
int snpgpu_multi_ibs_num(snpgpu_multi *m, int32_t *ibs0, int32_t *ibs1, int32_t *ibs2, int mem)
{
    if (need(m, SNPGPU_IBS, SNPGPU_IBS, "snpgpu_multi_ibs_num")) return 1;
    void *out[3] = {ibs0, ibs1, ibs2};
    return gather_slabs(m, 3, sizeof(int32_t), out, mem, [](snpgpu_ctx *c, void **p) {
        return snpgpu_ibs_num(c, (int32_t *)p[0], (int32_t *)p[1], (int32_t *)p[2], 1, SNPGPU_DEVICE);
    });
}

int snpgpu_multi_ibs_ave(snpgpu_multi *m, double *out_, int mem)
{
    if (need(m, SNPGPU_IBS, SNPGPU_IBS, "snpgpu_multi_ibs_ave")) return 1;
    void *out[1] = {out_};
    return gather_slabs(m, 1, sizeof(double), out, mem, [](snpgpu_ctx *c, void **p) { return snpgpu_ibs_ave(c, (double *)p[0], 1, SNPGPU_DEVICE); });
}

int snpgpu_multi_king_robust(snpgpu_multi *m, const int32_t *family, double *ibs0, double *kinship, int mem)
{
    if (need(m, SNPGPU_KING_ROBUST, SNPGPU_KING_ROBUST, "snpgpu_multi_king_robust")) return 1;
    void *out[2] = {ibs0, kinship};
    return gather_slabs(m, 2, sizeof(double), out, mem, [family](snpgpu_ctx *c, void **p) {
        return snpgpu_king_robust(c, family, (double *)p[0], (double *)p[1], 1, SNPGPU_DEVICE);
    });
}

int snpgpu_multi_king_robust_counts(snpgpu_multi *m, uint32_t *out5, int mem)
{
    if (need(m, SNPGPU_KING_ROBUST, SNPGPU_KING_ROBUST, "snpgpu_multi_king_robust_counts")) return 1;
    void *out[1] = {out5};
    return gather_slabs(m, 1, 5 * sizeof(uint32_t), out, mem, [](snpgpu_ctx *c, void **p) { return snpgpu_king_robust_counts(c, (uint32_t *)p[0], SNPGPU_DEVICE); });
}

int snpgpu_multi_king_homo(snpgpu_multi *m, double *k0, double *k1, int mem)
{
    if (need(m, SNPGPU_KING_HOMO, SNPGPU_KING_HOMO, "snpgpu_multi_king_homo")) return 1;
    void *out[2] = {k0, k1};
    return gather_slabs(m, 2, sizeof(double), out, mem, [](snpgpu_ctx *c, void **p) { return snpgpu_king_homo(c, (double *)p[0], (double *)p[1], 1, SNPGPU_DEVICE); });
}

int snpgpu_multi_grm_gcta(snpgpu_multi *m, double *out_, int mem)
{
    if (need(m, SNPGPU_GRM_GCTA, SNPGPU_GRM_GCTA, "snpgpu_multi_grm_gcta")) return 1;
    void *out[1] = {out_};
    return gather_slabs(m, 1, sizeof(double), out, mem, [](snpgpu_ctx *c, void **p) { return snpgpu_grm_gcta(c, (double *)p[0], 1, SNPGPU_DEVICE); });
}

int snpgpu_multi_eigmix(snpgpu_multi *m, int diagadj, double scale, double *out_, int mem)
{
    if (need(m, SNPGPU_EIGMIX, SNPGPU_EIGMIX, "snpgpu_multi_eigmix")) return 1;
    void *out[1] = {out_};
    return gather_slabs(m, 1, sizeof(double), out, mem, [=](snpgpu_ctx *c, void **p) { return snpgpu_eigmix(c, diagadj, scale, (double *)p[0], 1, SNPGPU_DEVICE); });
}

// trace of the whole matrix: the sum of the resident panels' diagonal parts (all panels of a one-pass plan)
int snpgpu_multi_pca_trace(snpgpu_multi *m, double *trace)
{
    if (need(m, SNPGPU_PCA_COV, SNPGPU_PCA_COV, "snpgpu_multi_pca_trace")) return 1;
    double tr = 0;
    for (snpgpu_ctx *c : m->ctx) {
        double t = 0;
        if (snpgpu_pca_panel_trace(c, &t)) return 1;
        tr += t;
    }
    if (trace) *trace = tr;
    return 0;
}

int snpgpu_multi_pca_cov(snpgpu_multi *m, double *out_, int normalize, double *trace_xtx, int mem)
{
    if (need(m, SNPGPU_PCA_COV, SNPGPU_PCA_COV, "snpgpu_multi_pca_cov")) return 1;
    double tr = 0;
    if (snpgpu_multi_pca_trace(m, &tr)) return 1;
    if (trace_xtx) *trace_xtx = tr;
    if (!out_) return 0;
    void *out[1] = {out_};
    return gather_slabs(m, 1, sizeof(double), out, mem, [=](snpgpu_ctx *c, void **p) {
        return snpgpu_pca_cov(c, (double *)p[0], 1, normalize, tr, nullptr, SNPGPU_DEVICE);
    });
}

int snpgpu_multi_finalize_inplace(snpgpu_multi *m, int diagadj, double scale)
{
    if (!m) { set_error("snpgpu_multi_finalize_inplace: NULL object"); return 1; }
    for (snpgpu_ctx *c : m->ctx)
        if (snpgpu_finalize_inplace(c, diagadj, scale)) return 1;
    return 0;
}

int snpgpu_multi_topk_eigen(snpgpu_multi *m, double scale, int k, const snpgpu_eig_opts *opts, double *eigval, double *eigvec,
                            int mem, snpgpu_eig_info *info)
{
    if (!m) { set_error("snpgpu_multi_topk_eigen: NULL object"); return 1; }
    if (opts && opts->reduce) { set_error("snpgpu_multi_topk_eigen: the object reduces over its own devices; no callback"); return 1; }
    // the resident panels must be the whole triangle
    int64_t rows = 0;
    for (snpgpu_ctx *c : m->ctx) rows += c->row1 - c->row0;
    if (rows != m->N) { set_error("snpgpu_multi_topk_eigen: needs all panels resident (a one-pass plan)"); return 1; }
    for (snpgpu_ctx *c : m->ctx)
        if (!(c->kind == SNPGPU_PCA_COV || c->frozen)) { set_error("snpgpu_multi_topk_eigen: call snpgpu_multi_finalize_inplace first"); return 1; }
    double sc = scale;
    if (m->kind == SNPGPU_PCA_COV && !(scale > 0)) {         // scale <= 0: the (n - 1) / trace factor of gnrPCA, src/genPCA.cpp:1386-1390
        double tr = 0;
        if (snpgpu_multi_pca_trace(m, &tr)) return 1;
        if (!(tr > 0) || !std::isfinite(tr)) { set_error("LAPACK::DSPEVX error (-1), infinite or missing values in the genetic covariance matrix!"); return 1; }
        sc = (double)(m->N - 1) / tr;
    }
    MultiOperator op(m, sc);
    std::vector<double> w((size_t)std::max(k, 1));
    const int rc = krylov_topk(op, k, opts, w.data(), eigvec, mem, info);
    for (snpgpu_ctx *c : m->ctx) {              // the solver's fp32 copies of the panels: memory back to the devices (eigen.hip)
        (void)hipSetDevice(c->device);
        c->acc_f32.release();
        c->acc_f32_valid = false;
    }
    if (rc) return 1;
    if (eigval) memcpy(eigval, w.data(), sizeof(double) * (size_t)k);
    return 0;
}

}  // extern "C"
