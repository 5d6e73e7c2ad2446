// Host side of alan_reduce: validation, canonicalisation, kernel choice, the PLATE two-stage plan.
#include <algorithm>
#include <cstring>

#include "plan.h"

namespace alan {

static int64_t iabs64(int64_t v) { return v < 0 ? -v : v; }

static bool valid_dtype(int32_t dt) { return dt == ALAN_F32 || dt == ALAN_F64; }
static size_t dtype_bytes(int32_t dt) { return dt == ALAN_F64 ? 8 : 4; }

int canonicalise(const alan_reduce_desc_t &d, uint32_t keep_mask, uint32_t red_mask,
                 const alan_tensor_t &out, Canon &c, uint32_t plate_mask, const alan_tensor_t *lse_out) {
    if (d.ndim < 0 || d.ndim > MAXD) return ALAN_ERR_BAD_DESC;
    if (d.n_factors < 1 || d.n_factors > MAXF) return ALAN_ERR_BAD_DESC;
    for (int i = 0; i < d.ndim; ++i)
        if (d.size[i] < 1) return ALAN_ERR_BAD_DESC;
    for (int f = 0; f < d.n_factors; ++f)
        if (!d.factor[f].data || !valid_dtype(d.factor[f].dtype)) return ALAN_ERR_BAD_DESC;
    if (!out.data || !valid_dtype(out.dtype)) return ALAN_ERR_BAD_DESC;
    const bool has_w = d.mode == ALAN_MODE_WEXPSUM;
    if (has_w && (!d.weight.data || !valid_dtype(d.weight.dtype))) return ALAN_ERR_BAD_DESC;

    // dominant factor: the one that spans the most elements of the space
    int dom = 0;
    double best = -1;
    for (int f = 0; f < d.n_factors; ++f) {
        double ext = 1;
        for (int i = 0; i < d.ndim; ++i)
            if (((keep_mask | red_mask) >> i) & 1)
                if (d.factor[f].stride[i] != 0) ext *= (double)d.size[i];
        if (ext > best) {
            best = ext;
            dom = f;
        }
    }
    c.dominant = dom;

    // order dims by the dominant factor's stride (outermost first); dims the dominant factor lacks
    // are ordered by the largest stride any other tensor gives them and go outermost.  When the
    // OUTPUT is the biggest tensor of the call (a producer: small inputs broadcast into a big
    // result) the keep dims follow the output's layout instead, so that the stores coalesce.
    double out_ext = 1;
    for (int i = 0; i < d.ndim; ++i)
        if ((keep_mask >> i) & 1) out_ext *= (double)d.size[i];
    const bool out_dominates = out_ext > best;
    auto sort_key = [&](int i) -> int64_t {
        if (out_dominates && ((keep_mask >> i) & 1) && out.stride[i] != 0) return iabs64(out.stride[i]);
        int64_t s = iabs64(d.factor[dom].stride[i]);
        if (s != 0) return s;
        int64_t alt = 0;
        for (int f = 0; f < d.n_factors; ++f) alt = std::max(alt, iabs64(d.factor[f].stride[i]));
        return (int64_t)1 << 62 | alt;
    };
    auto gather = [&](uint32_t mask, int *idx) {
        int n = 0;
        for (int i = 0; i < d.ndim; ++i)
            if (((mask >> i) & 1) && d.size[i] > 1) idx[n++] = i;
        std::stable_sort(idx, idx + n, [&](int a, int b) { return sort_key(a) > sort_key(b); });
        return n;
    };
    int kidx[MAXD], ridx[MAXD];
    int nk = gather(keep_mask, kidx);
    int nr = gather(red_mask, ridx);

    c.nf = d.n_factors;
    for (int f = 0; f < c.nf; ++f) {
        c.f[f].p = d.factor[f].data;
        c.f[f].dtype = d.factor[f].dtype;
        c.f[f].scale = d.factor[f].scale;
    }
    c.w.p = has_w ? d.weight.data : nullptr;
    c.w.dtype = d.weight.dtype;
    c.w.scale = 1.f;
    c.o.p = out.data;
    c.o.dtype = out.dtype;
    c.o.scale = 1.f;
    const bool has_l = lse_out != nullptr && lse_out->data != nullptr;
    c.l.p = has_l ? lse_out->data : nullptr;
    c.l.dtype = has_l ? lse_out->dtype : out.dtype;
    c.l.scale = 1.f;

    // merge adjacent dims that every tensor strides through contiguously
    auto build = [&](const int *idx, int n, bool keep, int64_t *size_out) -> int {
        int m = 0;
        int64_t fs[MAXF][MAXD], ws[MAXD], os[MAXD], ls[MAXD];
        bool pl[MAXD];
        for (int j = 0; j < n; ++j) {
            const int i = idx[j];
            bool merged = false;
            const bool is_plate = keep && ((plate_mask >> i) & 1);
            if (m > 0) {
                const int64_t sz = d.size[i];
                bool ok = pl[m - 1] == is_plate;
                if (ok && keep && has_l) ok = ls[m - 1] == lse_out->stride[i] * sz;
                for (int f = 0; f < c.nf && ok; ++f) ok = fs[f][m - 1] == d.factor[f].stride[i] * sz;
                if (ok && has_w) ok = ws[m - 1] == d.weight.stride[i] * sz;
                if (ok && keep) ok = os[m - 1] == out.stride[i] * sz;
                if (ok) {
                    size_out[m - 1] *= sz;
                    for (int f = 0; f < c.nf; ++f) fs[f][m - 1] = d.factor[f].stride[i];
                    ws[m - 1] = d.weight.stride[i];
                    os[m - 1] = out.stride[i];
                    if (has_l) ls[m - 1] = lse_out->stride[i];
                    merged = true;
                }
            }
            if (!merged) {
                size_out[m] = d.size[i];
                for (int f = 0; f < c.nf; ++f) fs[f][m] = d.factor[f].stride[i];
                ws[m] = has_w ? d.weight.stride[i] : 0;
                os[m] = keep ? out.stride[i] : 0;
                ls[m] = (keep && has_l) ? lse_out->stride[i] : 0;
                pl[m] = is_plate;
                ++m;
            }
        }
        for (int j = 0; j < m; ++j) {
            for (int f = 0; f < c.nf; ++f) (keep ? c.f[f].ks : c.f[f].rs)[j] = fs[f][j];
            (keep ? c.w.ks : c.w.rs)[j] = ws[j];
            if (keep) {
                c.o.ks[j] = os[j];
                c.l.ks[j] = ls[j];
                c.kplate[j] = pl[j];
            }
        }
        return m;
    };
    c.nk = build(kidx, nk, true, c.ksize);
    c.nr = build(ridx, nr, false, c.rsize);

    c.n_out = 1;
    c.n_red = 1;
    for (int j = 0; j < c.nk; ++j) c.n_out *= c.ksize[j];
    for (int j = 0; j < c.nr; ++j) c.n_red *= c.rsize[j];
    if (c.n_out >= (1ll << 31) || c.n_red >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    c.red_contig = c.nr > 0 && c.f[dom].rs[c.nr - 1] == 1;
    c.keep_contig = c.nk > 0 && c.f[dom].ks[c.nk - 1] == 1;
    return ALAN_OK;
}

int plan_group(const Canon &c, int out_dtype, double add_const, GroupDesc &gd, GroupLaunch &gl, float out_scale) {
    std::memset(&gd, 0, sizeof(gd));
    gd.nf = c.nf;
    gd.nk = c.nk;
    gd.nr = c.nr;
    gd.out_dtype = out_dtype;
    gd.n_out = (uint32_t)c.n_out;
    gd.n_red = (uint32_t)c.n_red;
    for (int j = 0; j < c.nk; ++j) gd.kdiv[j] = make_fastdiv((uint32_t)c.ksize[j]);
    for (int j = 0; j < c.nr; ++j) gd.rdiv[j] = make_fastdiv((uint32_t)c.rsize[j]);
    for (int f = 0; f < c.nf; ++f) gd.f[f] = c.f[f];
    gd.w = c.w;
    gd.out = const_cast<void *>(c.o.p);
    for (int j = 0; j < c.nk; ++j) gd.oks[j] = c.o.ks[j];
    gd.add_const = add_const;
    gd.out_scale = out_scale;
    if (gd.nr == 0) {  // the kernel's single-dim fast path reads rs[0]
        gd.nr = 1;
        gd.rdiv[0] = make_fastdiv(1);
    }

    // ---- lanes per output element
    const int64_t want_threads = 256ll * 256 * 4;  // >= 4 workgroups per CU
    int logG = 0;
    if (c.n_red > 1) {
        if (c.red_contig) {
            while (logG < 6 && (c.n_red >> logG) > 4) ++logG;  // ~4 elements per lane
        }
        // small launches are latency chains: widen the group until a lane sees one element
        while (logG < 6 && (c.n_out << logG) < want_threads && (1ll << logG) < c.n_red) ++logG;
    }
    gl.block = (logG == 6) && (c.n_out * 64 < want_threads) && (c.n_red >= 512);
    gl.logG = logG;
    if (gl.block) {
        gl.grid = (uint32_t)c.n_out;
    } else {
        const int64_t threads = c.n_out << logG;
        if (threads >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
        gl.grid = (uint32_t)((threads + 255) / 256);
    }
    return ALAN_OK;
}

// ------------------------------------------------------------------------------------------
// Layout of the per-(KEEP,PLATE) log-sum-exp values when the caller does not provide lse_out:
// contiguous, dims ordered like the largest factor stores them (so stage 1 writes coalesced).
static void plate_workspace_layout(const alan_reduce_desc_t &d, alan_tensor_t &v, int64_t &numel) {
    int dom = 0;
    double best = -1;
    for (int f = 0; f < d.n_factors; ++f) {
        double ext = 1;
        for (int i = 0; i < d.ndim; ++i)
            if (d.factor[f].stride[i] != 0) ext *= (double)d.size[i];
        if (ext > best) best = ext, dom = f;
    }
    int idx[MAXD], n = 0;
    for (int i = 0; i < d.ndim; ++i)
        if (d.role[i] != ALAN_REDUCE) idx[n++] = i;
    std::stable_sort(idx, idx + n, [&](int a, int b) {
        const int64_t sa = iabs64(d.factor[dom].stride[a]), sb = iabs64(d.factor[dom].stride[b]);
        return (sa == 0 ? (int64_t)1 << 62 : sa) > (sb == 0 ? (int64_t)1 << 62 : sb);
    });
    std::memset(&v, 0, sizeof(v));
    int64_t st = 1;
    for (int j = n - 1; j >= 0; --j) {
        v.stride[idx[j]] = d.size[idx[j]] > 1 ? st : 0;
        st *= d.size[idx[j]];
    }
    numel = st;
}

static int run_single(const alan_reduce_desc_t &d, uint32_t keep_mask, uint32_t red_mask, int mode,
                      const alan_tensor_t &out, double add_const, hipStream_t stream,
                      const EvPair &ev = EvPair()) {
    Canon c;
    int rc = canonicalise(d, keep_mask, red_mask, out, c);
    if (rc != ALAN_OK) return rc;
    const int compute = out.dtype;
    const bool producer = mode == ALAN_MODE_NORMAL || mode == ALAN_MODE_NORMAL_LOGSCALE || mode == ALAN_MODE_BERNOULLI ||
                          mode == ALAN_MODE_PRODUCER_GRAD;
    const float out_scale = producer ? out.scale : 1.f;
    if (ev.ring_n) {                     // result ring: the single-workgroup small kernel or nothing
        GroupDesc gd;
        GroupLaunch gl;
        rc = plan_group(c, out.dtype, add_const, gd, gl, out_scale);
        return rc != ALAN_OK ? rc : try_launch_small(c, gd, gl, mode, compute, stream, ev);
    }
    if (mode == ALAN_MODE_NORMAL || mode == ALAN_MODE_NORMAL_LOGSCALE) {
        rc = try_launch_normal_outer(c, mode == ALAN_MODE_NORMAL_LOGSCALE, out_scale, add_const, stream, ev);
        if (rc != ALAN_ERR_UNSUPPORTED) return rc;
    }
    const RowsPlan rp = plan_rows(c, mode, compute);
    if (rp.ok) return launch_rows(c, rp, mode, add_const, nullptr, 0, stream, ev);
    GroupDesc gd;
    GroupLaunch gl;
    rc = plan_group(c, out.dtype, add_const, gd, gl, out_scale);
    if (rc != ALAN_OK) return rc;
    rc = try_launch_small(c, gd, gl, mode, compute, stream, ev);
    if (rc != ALAN_ERR_UNSUPPORTED) return rc;
    return launch_group(gd, gl, mode, compute, stream, ev);
}

// The fused plan for "log-sum-exp over REDUCE then sum over PLATE", if the rows kernel can take it.
static bool plan_fused_plate(const alan_reduce_desc_t &d, uint32_t keep, uint32_t red, uint32_t plate,
                             Canon &c, RowsPlan &rp) {
    if (canonicalise(d, keep | plate, red, d.out, c, plate, &d.lse_out) != ALAN_OK) return false;
    rp = plan_rows(c, ALAN_MODE_LSE, d.out.dtype);
    return rp.ok;
}

static int classify(const alan_reduce_desc_t &d, uint32_t &keep, uint32_t &red, uint32_t &plate) {
    keep = red = plate = 0;
    if (d.ndim < 0 || d.ndim > MAXD) return ALAN_ERR_BAD_DESC;
    for (int i = 0; i < d.ndim; ++i) {
        switch (d.role[i]) {
            case ALAN_KEEP: keep |= 1u << i; break;
            case ALAN_REDUCE: red |= 1u << i; break;
            case ALAN_PLATE: plate |= 1u << i; break;
            default: return ALAN_ERR_BAD_DESC;
        }
    }
    if (plate && d.mode != ALAN_MODE_LSE) return ALAN_ERR_BAD_DESC;
    if (d.mode != ALAN_MODE_LSE && d.mode != ALAN_MODE_SUM && d.mode != ALAN_MODE_WEXPSUM &&
        d.mode != ALAN_MODE_NORMAL && d.mode != ALAN_MODE_BERNOULLI && d.mode != ALAN_MODE_NORMAL_LOGSCALE &&
        d.mode != ALAN_MODE_PRODUCER_GRAD && d.mode != ALAN_MODE_DOT && d.mode != ALAN_MODE_AFFINE)
        return ALAN_ERR_BAD_DESC;
    if (d.mode == ALAN_MODE_DOT && (d.n_factors < 2 || d.n_factors > 5)) return ALAN_ERR_BAD_DESC;
    if (d.mode == ALAN_MODE_AFFINE && d.n_factors != 3) return ALAN_ERR_BAD_DESC;
    if (d.mode == ALAN_MODE_NORMAL_LOGSCALE && d.n_factors != 3) return ALAN_ERR_BAD_DESC;
    if (d.mode == ALAN_MODE_NORMAL && d.n_factors != 3 && d.n_factors != 6) return ALAN_ERR_BAD_DESC;
    if (d.mode == ALAN_MODE_BERNOULLI && d.n_factors != 2) return ALAN_ERR_BAD_DESC;
    if (d.mode == ALAN_MODE_PRODUCER_GRAD) {
        const int kind = (int)d.factor[0].scale;
        if (kind < 1 || kind > 4 || d.n_factors != (kind == 4 ? 3 : 4)) return ALAN_ERR_BAD_DESC;
    }
    return ALAN_OK;
}

// A log-sum-exp / sum with FEW outputs over a HUGE reduce space of several dims (a whole factor reduced to a handful of
// values) would be one workgroup per output walking millions of strided elements.  Two launches instead: the largest
// reduce dim is kept by the first (outputs x that dim: plenty of parallelism, short reductions), reduced by the second.
// (log-sum-exp of log-sum-exps is the log-sum-exp; the reference's +eps enters once per stage: 1e-7 relative.)
static int peel_dim(const alan_reduce_desc_t &d, uint32_t keep, uint32_t red, uint32_t plate) {
    if (plate || (d.mode != ALAN_MODE_LSE && d.mode != ALAN_MODE_SUM) || d.lse_out.data || d.ev_start || d.ev_stop) return -1;
    double n_out = 1, n_red = 1;
    int nred = 0, best = -1;
    for (int i = 0; i < d.ndim; ++i) {
        if (((keep >> i) & 1)) n_out *= (double)d.size[i];
        if (((red >> i) & 1) && d.size[i] > 1) {
            n_red *= (double)d.size[i];
            ++nred;
            if (best < 0 || d.size[i] > d.size[best]) best = i;
        }
    }
    if (nred < 2) return -1;
    if (!(n_out <= 256 && n_red >= 65536) && !(n_out <= 2048 && n_red >= 4096 && n_out * n_red >= (double)(1 << 21))) return -1;
    if (n_out * (double)d.size[best] > (double)(1 << 24)) return -1;
    return best;
}

static void peel_layout(const alan_reduce_desc_t &d, uint32_t keep, int p, alan_tensor_t &v, int64_t &numel) {
    std::memset(&v, 0, sizeof(v));
    int64_t st = 1;
    for (int i = d.ndim - 1; i >= 0; --i)
        if ((keep >> i) & 1) {
            v.stride[i] = d.size[i] > 1 ? st : 0;
            st *= d.size[i];
        }
    v.stride[p] = st;
    numel = st * d.size[p];
    v.dtype = d.out.dtype;
    v.scale = 1.f;
}

// The same situation with ONE long reduce dim (a column sum of a tall matrix): view the dim as [C, N / C] (a free
// descriptor slot permitting, N having a divisor of the right size) and let peel_dim() take it from there.
static bool split_long_dim(const alan_reduce_desc_t &d, uint32_t keep, uint32_t red, uint32_t plate,
                           alan_reduce_desc_t &out) {
    if (plate || (d.mode != ALAN_MODE_LSE && d.mode != ALAN_MODE_SUM) || d.lse_out.data || d.ev_start || d.ev_stop) return false;
    if (d.ndim >= MAXD) return false;
    double n_out = 1;
    int p = -1, nred = 0;
    for (int i = 0; i < d.ndim; ++i) {
        if ((keep >> i) & 1) n_out *= (double)d.size[i];
        if (((red >> i) & 1) && d.size[i] > 1) p = i, ++nred;
    }
    // few outputs over one long dim: up to 2048 outputs when the dim is >= 4096 long and the problem >= 2 Mi elements
    if (nred != 1) return false;
    const bool tall = n_out <= 256 && d.size[p] >= 65536;
    const bool wide = n_out <= 2048 && d.size[p] >= 4096 && n_out * (double)d.size[p] >= (double)(1 << 21);
    if (!tall && !wide) return false;
    const int64_t N = d.size[p];
    int64_t C = 0;
    for (int64_t c = std::min<int64_t>(1024, N / 64); c >= 16; --c)
        if (N % c == 0) {
            C = c;
            break;
        }
    if (!C) return false;
    out = d;
    const int q = d.ndim;
    out.ndim = d.ndim + 1;
    out.size[q] = C;
    out.role[q] = ALAN_REDUCE;
    out.size[p] = N / C;
    for (int f = 0; f < d.n_factors; ++f) out.factor[f].stride[q] = d.factor[f].stride[p] * (N / C);
    out.out.stride[q] = 0;
    return true;
}

// alan_reduce_batch: is this problem one launch of the small kernel?  Fills what that launch needs.
// ALAN_MODE_BERNOULLI_LINEAR -> the kernel's argument and launch geometry.  ALAN_ERR_UNSUPPORTED: a well-formed problem
// outside what the kernel takes (the caller evaluates the logits itself and uses ALAN_MODE_BERNOULLI).
static int lin_prepare(const alan_reduce_desc_t &d, LinDesc &ld, GroupLaunch &gl) {
    const bool grad = d.mode == ALAN_MODE_BERNOULLI_LINEAR_GRAD;
    if (d.ndim < 0 || d.ndim > MAXD || d.n_factors < 2 || d.n_factors > MAXF) return ALAN_ERR_BAD_DESC;
    if (d.lse_out.data || (d.weight.data != nullptr) != grad || d.ring_n || !d.out.data) return ALAN_ERR_BAD_DESC;
    if (grad && d.weight.dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
    for (int f = 0; f < d.n_factors; ++f) {
        if (!d.factor[f].data) return ALAN_ERR_BAD_DESC;
        if (d.factor[f].dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
    }
    if (d.out.dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
    std::memset(&ld, 0, sizeof(ld));
    int keep[MAXD], red[MAXD], nk = 0, nr = 0;
    uint32_t dot = 0;
    int64_t n_out = 1, n_red = 1;
    for (int i = 0; i < d.ndim; ++i) {
        if (d.size[i] < 1) return ALAN_ERR_BAD_DESC;
        switch (d.role[i]) {
            case ALAN_KEEP:
                if (d.size[i] > 1) keep[nk++] = i;
                n_out *= d.size[i];
                break;
            case ALAN_REDUCE:
                if (d.size[i] > 1) red[nr++] = i;
                n_red *= d.size[i];
                break;
            case ALAN_DOT: dot |= 1u << i; break;
            default: return ALAN_ERR_BAD_DESC;
        }
        if (n_out >= (1ll << 31) || n_red >= (1ll << 31) || d.size[i] >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    }
    if (nk > LIN_NK || nr > LIN_NR) return ALAN_ERR_UNSUPPORTED;
    // threads run along the output's innermost dim; lanes of a group along the value's
    // (the gradient's layout is that of `a`: its innermost keep dim sits above the dot dim)
    std::sort(keep, keep + nk, [&](int a, int b) { return d.out.stride[a] > d.out.stride[b]; });
    std::sort(red, red + nr, [&](int a, int b) { return d.factor[0].stride[a] > d.factor[0].stride[b]; });
    const int ko = LIN_NK - nk, ro = LIN_NR - nr;
    for (int k = 0; k < LIN_NK; ++k) ld.kdiv[k] = make_fastdiv(k < ko ? 1u : (uint32_t)d.size[keep[k - ko]]);
    for (int k = 0; k < LIN_NR; ++k) ld.rdiv[k] = make_fastdiv(k < ro ? 1u : (uint32_t)d.size[red[k - ro]]);
    const int64_t lim = (1ll << 31) - 1;
    bool fits = true;
    // strides of one tensor over the keep / summed dims, with the int32 reach check of the kernel's offsets
    auto lay = [&](const alan_tensor_t &x, int32_t *ks, int32_t *rs, int dot_dim) {
        int64_t reach = 0;
        for (int j = 0; j < nk; ++j) {
            const int64_t st = x.stride[keep[j]];
            reach += (d.size[keep[j]] - 1) * (st < 0 ? -st : st);
            if (st > lim || st < -lim) fits = false; else ks[ko + j] = (int32_t)st;
        }
        for (int j = 0; rs && j < nr; ++j) {
            const int64_t st = x.stride[red[j]];
            reach += (d.size[red[j]] - 1) * (st < 0 ? -st : st);
            if (st > lim || st < -lim) fits = false; else rs[ro + j] = (int32_t)st;
        }
        if (dot_dim >= 0) {
            const int64_t st = x.stride[dot_dim];
            reach += (d.size[dot_dim] - 1) * (st < 0 ? -st : st);
            if (st > lim || st < -lim) fits = false;
        }
        if (reach > lim) fits = false;
    };
    auto dot_strides = [&](const alan_tensor_t &x) {
        uint32_t m = 0;
        for (int i = 0; i < d.ndim; ++i)
            if (((dot >> i) & 1) && x.stride[i] != 0 && d.size[i] > 1) m |= 1u << i;
        return m;
    };
    if (dot_strides(d.factor[0])) return ALAN_ERR_BAD_DESC;
    lay(d.factor[0], ld.vks, ld.vrs, -1);
    for (int j = 0; j < nr; ++j)
        if (d.out.stride[red[j]] != 0) return ALAN_ERR_BAD_DESC;
    if (grad) {
        // the upstream gradient over the keep dims; the gradient's own dot stride (out is laid out like `a`)
        for (int j = 0; j < nr; ++j)
            if (d.weight.stride[red[j]] != 0) return ALAN_ERR_BAD_DESC;
        if (dot_strides(d.weight)) return ALAN_ERR_BAD_DESC;
        lay(d.weight, ld.gks, nullptr, -1);
        ld.g = (const float *)d.weight.data;
        if (dot == 0 || (dot & (dot - 1))) return ALAN_ERR_UNSUPPORTED;        // (exactly one DOT dim: the first term's)
        int dd = -1;
        for (int i = 0; i < d.ndim; ++i)
            if ((dot >> i) & 1) dd = i;
        lay(d.out, ld.oks, nullptr, dd);
        if (d.out.stride[dd] > lim || d.out.stride[dd] < -lim) return ALAN_ERR_UNSUPPORTED;
        ld.ods = (int32_t)d.out.stride[dd];
    } else {
        lay(d.out, ld.oks, nullptr, -1);
    }
    ld.val = (const float *)d.factor[0].data;
    ld.out = (float *)d.out.data;
    // terms
    int nt = 0, f = 1;
    while (f < d.n_factors) {
        if ((int)d.factor[f].scale != nt + 1) return ALAN_ERR_BAD_DESC;
        if (nt == LIN_T) return ALAN_ERR_UNSUPPORTED;
        const alan_tensor_t &a = d.factor[f];
        const bool two = f + 1 < d.n_factors && (int)d.factor[f + 1].scale == nt + 1;
        ld.a[nt] = (const float *)a.data;
        if (!two) {
            if (dot_strides(a)) return ALAN_ERR_BAD_DESC;
            lay(a, ld.aks[nt], ld.ars[nt], -1);
            ld.len[nt] = 1;
            f += 1;
        } else {
            const alan_tensor_t &b = d.factor[f + 1];
            const uint32_t m = dot_strides(a) | dot_strides(b);
            if (m & (m - 1)) return ALAN_ERR_UNSUPPORTED;          // contracted over more than one dim
            int dd = -1;
            for (int i = 0; i < d.ndim; ++i)
                if ((m >> i) & 1) dd = i;
            lay(a, ld.aks[nt], ld.ars[nt], dd);
            lay(b, ld.bks[nt], ld.brs[nt], dd);
            ld.b[nt] = (const float *)b.data;
            ld.len[nt] = dd >= 0 ? (int32_t)d.size[dd] : 1;
            if (fits && dd >= 0) {
                ld.ads[nt] = (int32_t)a.stride[dd];
                ld.bds[nt] = (int32_t)b.stride[dd];
            }
            f += 2;
        }
        ++nt;
    }
    if (!fits) return ALAN_ERR_UNSUPPORTED;
    ld.nt = nt;
    ld.n_out = (uint32_t)n_out;
    ld.n_red = (uint32_t)n_red;
    ld.out_scale = d.out.scale;
    ld.add_const = (float)d.add_const;
    if (grad) {
        // the gradient kernel: a thread per row of `a` -- term 0 must be a dot product of at most 32 events whose first
        // operand carries every keep dim and no summed dim
        if (nt < 1 || !ld.b[0] || ld.len[0] > 32 || ld.len[0] < 1) return ALAN_ERR_UNSUPPORTED;
        for (int k = 0; k < LIN_NR; ++k)
            if (ld.ars[0][k] != 0) return ALAN_ERR_UNSUPPORTED;
        for (int j = 0; j < nk; ++j)
            if (ld.aks[0][ko + j] == 0) return ALAN_ERR_UNSUPPORTED;
        for (int tm = 1; tm < nt; ++tm)
            if (ld.b[tm] && ld.len[tm] > 8 * 64) return ALAN_ERR_UNSUPPORTED;
        gl.logG = 0, gl.block = false, gl.grid = (uint32_t)((n_out + 255) / 256);
        return ALAN_OK;
    }
    // lanes per output element: as few as fill the chip (~160 k threads) -- a lane's prologue, the index decomposition
    // of its output, is ~100 instructions, and at 64 lanes per output a lane of bus_breakdown's 150-observation plate paid
    // it for 2 or 3 elements (K = 100, 60 k outputs: 57 us at 64 lanes, 32 us at 4; K = 30, 5.4 k outputs: 8.6 us at 32
    // lanes, 13.6 at 8)
    int logG = 0;
    while (logG < 6 && (n_out << logG) < 160000 && (1ll << logG) < n_red) ++logG;
    gl.block = logG == 6 && n_red >= 512 && n_out * 64 < 256ll * 256 * 4;
    gl.logG = logG;
    if (gl.block) {
        gl.grid = (uint32_t)n_out;
    } else {
        const int64_t threads = n_out << logG;
        if (threads >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
        gl.grid = (uint32_t)((threads + 255) / 256);
    }
    return ALAN_OK;
}

// LSE over REDUCE then sum over PLATE as one small launch (SmallPlateDesc), or false when the problem is not small /
// not fp32 / has more dims than the kernel walks.
static bool small_plate_prepare(const alan_reduce_desc_t &d, SmallPlateDesc &sd, GroupLaunch &gl) {
    if (d.mode != ALAN_MODE_LSE || d.ndim < 0 || d.ndim > MAXD || d.n_factors < 1 || d.n_factors > MAXF) return false;
    if (d.weight.data || d.ring_n || !d.out.data || d.out.dtype != ALAN_F32) return false;
    if (d.lse_out.data && d.lse_out.dtype != ALAN_F32) return false;
    for (int f = 0; f < d.n_factors; ++f)
        if (!d.factor[f].data || d.factor[f].dtype != ALAN_F32) return false;
    std::memset(&sd, 0, sizeof(sd));
    int keep[MAXD], pl[MAXD], red[MAXD], nk = 0, np = 0, nr = 0;
    int64_t n_out = 1, n_plate = 1, n_red = 1;
    for (int i = 0; i < d.ndim; ++i) {
        if (d.size[i] < 1) return false;
        if (d.size[i] == 1) continue;
        switch (d.role[i]) {
            case ALAN_KEEP: keep[nk++] = i, n_out *= d.size[i]; break;
            case ALAN_PLATE: pl[np++] = i, n_plate *= d.size[i]; break;
            case ALAN_REDUCE: red[nr++] = i, n_red *= d.size[i]; break;
            default: return false;
        }
        if (n_out * n_plate * n_red > (1ll << 22)) return false;        // bigger ones: the rows kernel / two launches
    }
    if (nk > SP_NK || np > SP_NP || nr > SP_NR || np == 0 || nr == 0 || n_plate > 4096) return false;
    std::sort(keep, keep + nk, [&](int a, int b) { return d.out.stride[a] > d.out.stride[b]; });
    int dom = 0;                                          // lanes run along the largest factor's innermost reduce dim
    for (int f = 1; f < d.n_factors; ++f) {
        auto ext = [&](int g) {
            int64_t e = 1;
            for (int i = 0; i < d.ndim; ++i)
                if (d.factor[g].stride[i] != 0) e *= d.size[i];
            return e;
        };
        if (ext(f) > ext(dom)) dom = f;
    }
    std::sort(red, red + nr, [&](int a, int b) { return d.factor[dom].stride[a] > d.factor[dom].stride[b]; });
    const int ko = SP_NK - nk, po = SP_NP - np, ro = SP_NR - nr;
    for (int k = 0; k < SP_NK; ++k) sd.kdiv[k] = make_fastdiv(k < ko ? 1u : (uint32_t)d.size[keep[k - ko]]);
    for (int k = 0; k < SP_NP; ++k) sd.pdiv[k] = make_fastdiv(k < po ? 1u : (uint32_t)d.size[pl[k - po]]);
    for (int k = 0; k < SP_NR; ++k) sd.rdiv[k] = make_fastdiv(k < ro ? 1u : (uint32_t)d.size[red[k - ro]]);
    const int64_t lim = (1ll << 31) - 1;
    bool fits = true;
    auto lay = [&](const alan_tensor_t &x, int32_t *ks, int32_t *ps, int32_t *rs) {
        int64_t reach = 0;
        auto put = [&](const int *dims, int n, int off, int32_t *dst) {
            for (int j = 0; j < n; ++j) {
                const int64_t st = x.stride[dims[j]];
                reach += (d.size[dims[j]] - 1) * (st < 0 ? -st : st);
                if (st > lim || st < -lim) fits = false; else if (dst) dst[off + j] = (int32_t)st;
            }
        };
        put(keep, nk, ko, ks);
        put(pl, np, po, ps);
        put(red, nr, ro, rs);
        if (reach > lim) fits = false;
    };
    for (int f = 0; f < MAXF; ++f) {
        const alan_tensor_t &src = d.factor[f < d.n_factors ? f : 0];
        sd.f[f] = (const float *)src.data;
        sd.fscale[f] = f < d.n_factors ? src.scale : 0.f;
        if (f < d.n_factors) lay(src, sd.fks[f], sd.fps[f], sd.frs[f]);
    }
    for (int j = 0; j < np; ++j)
        if (d.out.stride[pl[j]] != 0) return false;
    for (int j = 0; j < nr; ++j)
        if (d.out.stride[red[j]] != 0 || (d.lse_out.data && d.lse_out.stride[red[j]] != 0)) return false;
    lay(d.out, sd.oks, nullptr, nullptr);
    if (d.lse_out.data) lay(d.lse_out, sd.lks, sd.lps, nullptr);
    if (!fits) return false;
    sd.out = (float *)d.out.data;
    sd.lse = (float *)d.lse_out.data;
    sd.n_out = (uint32_t)n_out, sd.n_plate = (uint32_t)n_plate, sd.n_red = (uint32_t)n_red;
    sd.nf = d.n_factors;
    sd.add_const = (float)d.add_const;
    int logG = 0;
    while (logG < 6 && (1ll << logG) < n_red) ++logG;
    gl.block = false;
    gl.logG = logG;
    const int64_t threads = n_out << logG;
    if (threads >= (1ll << 31)) return false;
    gl.grid = (uint32_t)((threads + 255) / 256);
    return true;
}

// ALAN_MODE_NORMAL_TABLE: one workgroup of the small-problem launch (normal_lse_table.h).  Not a reduction: the descriptor
// only names the scale rows (one KEEP dim) and the event (one REDUCE dim) of the single factor.
static int table_prepare(const alan_reduce_desc_t &d, SmallDesc &sd, GroupLaunch &gl) {
    if (d.ndim < 0 || d.ndim > MAXD || d.n_factors != 1 || !d.factor[0].data || !d.out.data) return ALAN_ERR_BAD_DESC;
    if (d.weight.data || d.lse_out.data || d.ring_n || d.noise.on || d.ev_start || d.ev_stop) return ALAN_ERR_BAD_DESC;
    if (d.factor[0].dtype != ALAN_F32 || d.out.dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
    if ((uintptr_t)d.out.data & 15) return ALAN_ERR_BAD_DESC;
    int64_t ns = 1, e = 1, s_ss = 0, s_se = 0;
    int nkd = 0, nrd = 0;
    for (int i = 0; i < d.ndim; ++i) {
        if (d.size[i] < 1) return ALAN_ERR_BAD_DESC;
        if (d.size[i] == 1) continue;
        if (d.role[i] == ALAN_KEEP) ns = d.size[i], s_ss = d.factor[0].stride[i], ++nkd;
        else if (d.role[i] == ALAN_REDUCE) e = d.size[i], s_se = d.factor[0].stride[i], ++nrd;
        else return ALAN_ERR_BAD_DESC;
    }
    if (nkd > 1 || nrd > 1) return ALAN_ERR_BAD_DESC;
    if (ns > 32 || e > 32 || s_ss < 0 || s_se < 0 || ns * s_ss + e * s_se >= (1ll << 31)) return ALAN_ERR_UNSUPPORTED;
    std::memset(&sd, 0, sizeof(sd));
    sd.f[0] = (const float *)d.factor[0].data;
    sd.out = (float *)d.out.data;
    sd.n_out = (uint32_t)ns, sd.n_red = (uint32_t)e, sd.nf = 1;
    sd.fks[0][SMALL_NK - 1] = (int32_t)s_ss, sd.frs[0][SMALL_NR - 1] = (int32_t)s_se;
    sd.fscale[0] = d.factor[0].scale;
    gl = GroupLaunch();
    gl.grid = 1;
    return ALAN_OK;
}

static bool prepare_small(const alan_reduce_desc_t &d, SmallDesc &sd, GroupLaunch &gl, int &mode) {
    if (d.mode == ALAN_MODE_NORMAL_TABLE) {
        mode = d.mode;
        return table_prepare(d, sd, gl) == ALAN_OK;
    }
    uint32_t keep, red, plate;
    if (classify(d, keep, red, plate) != ALAN_OK || plate || d.ev_start || d.ev_stop) return false;
    if (peel_dim(d, keep, red, plate) >= 0) return false;
    {
        alan_reduce_desc_t d2;
        if (split_long_dim(d, keep, red, plate, d2)) return false;
    }
    mode = d.mode;
    if (mode == ALAN_MODE_LSE && red == 0) mode = ALAN_MODE_SUM;
    Canon c;
    if (canonicalise(d, keep, red, d.out, c) != ALAN_OK) return false;
    const bool producer = mode == ALAN_MODE_NORMAL || mode == ALAN_MODE_NORMAL_LOGSCALE || mode == ALAN_MODE_BERNOULLI ||
                          mode == ALAN_MODE_PRODUCER_GRAD;
    const float out_scale = producer ? d.out.scale : 1.f;
    if ((mode == ALAN_MODE_NORMAL || mode == ALAN_MODE_NORMAL_LOGSCALE) &&
        try_launch_normal_outer(c, mode == ALAN_MODE_NORMAL_LOGSCALE, out_scale, d.add_const, nullptr, EvPair(), true) !=
            ALAN_ERR_UNSUPPORTED)
        return false;
    if (plan_rows(c, mode, d.out.dtype).ok) return false;
    GroupDesc gd;
    if (plan_group(c, d.out.dtype, d.add_const, gd, gl, out_scale) != ALAN_OK) return false;
    if (build_small(c, gd, mode, d.out.dtype, sd) != ALAN_OK) return false;
    if (gd.n_out == 0) gl.grid = 0;
    return true;
}

}  // namespace alan

using namespace alan;

extern "C" int alan_reduce_batch(const alan_reduce_desc_t *const *descs, int32_t n, void *stream_) {
    if (!descs || n < 0) return ALAN_ERR_BAD_DESC;
    for (int i = 0; i < n; ++i)
        if (!descs[i]) return ALAN_ERR_BAD_DESC;
    hipStream_t stream = (hipStream_t)stream_;
    // the problems are independent, so their order is free: every small one first, SMALL_MULTI per launch, then the
    // others one by one
    SmallDesc sd[SMALL_MULTI];
    GroupLaunch gl[SMALL_MULTI];
    LinDesc lin;
    bool have_lin = false;
    int mode[SMALL_MULTI], m = 0, lone = -1;
    std::memset(sd, 0, sizeof(sd));
    if (n > 64) return ALAN_ERR_UNSUPPORTED;
    // generated noise (alan_noise_t): every such problem must be of the small kind (found out before anything is launched),
    // all of them name the same generator, and the last launch that holds one advances its counter
    const alan_noise_t *nz = nullptr, *carry = nullptr;
    int last_noise = -1;
    bool grp_noise = false, grp_last = false;
    for (int i = 0; i < n; ++i) {
        const alan_noise_t &z = descs[i]->noise;
        if (!z.on) continue;
        if (z.on == 2) {                 // a hand-on carried by this batch's first small-problem launch (alan_noise_t)
            carry = &z;
            continue;
        }
        SmallDesc ts;
        GroupLaunch tg;
        int tm;
        if (descs[i]->mode != ALAN_MODE_AFFINE && descs[i]->mode != ALAN_MODE_DOT) return ALAN_ERR_BAD_DESC;
        if (!prepare_small(*descs[i], ts, tg, tm)) return ALAN_ERR_UNSUPPORTED;
        if (nz && (nz->seed != z.seed || nz->cell != z.cell || nz->receipt != z.receipt || nz->advance != z.advance ||
                   nz->advance_by != z.advance_by))
            return ALAN_ERR_BAD_DESC;
        nz = &z, last_noise = i;
    }
    if (carry) {
        if (nz) return ALAN_ERR_BAD_DESC;                // (not beside problems that draw: they hand on themselves)
        bool any_small = false;
        for (int i = 0; i < n && !any_small; ++i) {
            SmallDesc ts;
            GroupLaunch tg;
            int tm;
            any_small = descs[i]->mode != ALAN_MODE_BERNOULLI_LINEAR && prepare_small(*descs[i], ts, tg, tm);
        }
        if (!any_small) return ALAN_ERR_UNSUPPORTED;     // (nothing launched)
    }
    auto flush_small = [&]() -> int {
        int rc = ALAN_OK;
        if (carry && m >= 1) {
            alan_noise_t c = *carry;                     // {cell -> *advance}: advance_by = 0, no receipt
            c.on = 1, c.advance_by = 0, c.receipt = nullptr;
            rc = launch_small_multi(sd, gl, mode, m, stream, have_lin ? &lin : nullptr, &c, true);
            carry = nullptr;
        } else if (m >= 2 || (m == 1 && grp_noise))
            rc = launch_small_multi(sd, gl, mode, m, stream, have_lin ? &lin : nullptr, grp_noise ? nz : nullptr, grp_last);
        else if (m == 1)
            rc = alan_reduce(descs[lone], nullptr, 0, stream_);
        m = 0;
        have_lin = false;
        grp_noise = grp_last = false;
        return rc;
    };
    bool other[64];
    for (int i = 0; i < n; ++i) {
        if (descs[i]->mode == ALAN_MODE_BERNOULLI_LINEAR) {
            // one per multi launch (its argument has its own slot there); a second one goes out alone, below
            bool tiles = false;                       // (plain terms, millions of elements: the tile kernel, a launch of its own)
            {
                uint32_t keep = 0, red = 0;
                for (int k = 0; k < descs[i]->ndim && k < MAXD; ++k) {
                    if (descs[i]->role[k] == ALAN_KEEP) keep |= 1u << k;
                    if (descs[i]->role[k] == ALAN_REDUCE) red |= 1u << k;
                }
                PairDesc pd;
                dim3 grid;
                size_t lds;
                tiles = descs[i]->ndim <= MAXD && pair_prepare(*descs[i], keep, red, 0, pd, grid, lds);
            }
            other[i] = tiles || have_lin || descs[i]->ev_start || descs[i]->ev_stop || lin_prepare(*descs[i], lin, gl[m]) != ALAN_OK;
            if (!other[i]) {
                have_lin = true;
                mode[m] = ALAN_MODE_BERNOULLI_LINEAR;
            }
        } else {
            other[i] = !prepare_small(*descs[i], sd[m], gl[m], mode[m]);
            if (!other[i] && descs[i]->noise.on == 1) {
                sd[m].noise_on = 1, sd[m].noise_off = descs[i]->noise.offset;
                grp_noise = true;
                if (i == last_noise) grp_last = true;
            }
        }
        if (other[i]) continue;
        lone = i;
        if (++m == SMALL_MULTI) {
            const int rc = flush_small();
            if (rc != ALAN_OK) return rc;
        }
    }
    int rc = flush_small();
    if (rc != ALAN_OK) return rc;
    for (int i = 0; i < n; ++i) {
        if (!other[i]) continue;
        if (alan_reduce_workspace_bytes(descs[i]) != 0) return ALAN_ERR_WORKSPACE;
        rc = alan_reduce(descs[i], nullptr, 0, stream_);
        if (rc != ALAN_OK) return rc;
    }
    return ALAN_OK;
}

// Role ALAN_PRESUM: the descriptor with that dim taken out (size 1) and the factor that carries it moved to slot 0,
// which the small kernel then reads as the sum of `n` slices `stride` apart.  -1: no such dim; -2: malformed.
static int strip_presum(const alan_reduce_desc_t &d, alan_reduce_desc_t &d2, int64_t &n, int64_t &stride) {
    int p = -1;
    for (int i = 0; i < d.ndim && i < MAXD; ++i)
        if (d.role[i] == ALAN_PRESUM) {
            if (p >= 0) return -2;
            p = i;
        }
    if (p < 0) return -1;
    if (d.mode != ALAN_MODE_LSE && d.mode != ALAN_MODE_SUM) return -2;
    if (d.n_factors < 1 || d.n_factors > MAXF) return -2;
    int f = -1;
    for (int i = 0; i < d.n_factors; ++i)
        if (d.factor[i].stride[p] != 0) {
            if (f >= 0) return -2;
            f = i;
        }
    if (d.out.stride[p] != 0 || d.weight.data || d.lse_out.data) return -2;
    for (int i = 0; i < d.ndim; ++i)
        if (d.role[i] == ALAN_PLATE) return -2;
    d2 = d;
    n = d.size[p];
    stride = f >= 0 ? d.factor[f].stride[p] : 0;
    d2.size[p] = 1;
    d2.role[p] = ALAN_KEEP;
    if (f > 0) std::swap(d2.factor[0], d2.factor[f]);
    if (f < 0 || n < 1) return -2;          // (a dim nobody carries cannot be summed: its size would multiply the result)
    return p;
}

// The single small-kernel launch a PRESUM problem is (or ALAN_ERR_UNSUPPORTED); dry: check only.
static int run_presum(const alan_reduce_desc_t &d2, int64_t n, int64_t stride, hipStream_t stream, bool dry) {
    uint32_t keep, red, plate;
    int rc = classify(d2, keep, red, plate);
    if (rc != ALAN_OK) return rc;
    if (d2.out.dtype != ALAN_F32) return ALAN_ERR_UNSUPPORTED;
    EvPair ev;
    ev.start = (hipEvent_t)d2.ev_start;
    ev.stop = (hipEvent_t)d2.ev_stop;
    if (d2.ring_n) {
        if (d2.ring_n < 0 || !d2.ring_slots || !d2.ring_counter) return ALAN_ERR_BAD_DESC;
        ev.ring_slots = d2.ring_slots, ev.ring_counter = d2.ring_counter, ev.ring_n = d2.ring_n, ev.ring_and_out = d2.ring_and_out;
    }
    const int mode = (d2.mode == ALAN_MODE_LSE && red == 0) ? ALAN_MODE_SUM : d2.mode;     // (log-sum-exp over no dims)
    Canon c;
    rc = canonicalise(d2, keep, red, d2.out, c);
    if (rc != ALAN_OK) return rc;
    GroupDesc gd;
    GroupLaunch gl;
    rc = plan_group(c, d2.out.dtype, d2.add_const, gd, gl, 1.f);
    if (rc != ALAN_OK) return rc;
    return try_launch_small(c, gd, gl, mode, d2.out.dtype, stream, ev, n, stride, dry);
}

extern "C" int alan_reduce_check(const alan_reduce_desc_t *d) {
    if (!d) return ALAN_ERR_BAD_DESC;
    {
        alan_reduce_desc_t d2;
        int64_t n, stride;
        const int p = strip_presum(*d, d2, n, stride);
        if (p == -2) return ALAN_ERR_BAD_DESC;
        if (p >= 0) return run_presum(d2, n, stride, nullptr, true);
    }
    if (d->mode == ALAN_MODE_NORMAL_TABLE) {
        SmallDesc sd;
        GroupLaunch gl;
        return table_prepare(*d, sd, gl);
    }
    if (d->mode == ALAN_MODE_BERNOULLI_LINEAR || d->mode == ALAN_MODE_BERNOULLI_LINEAR_GRAD) {
        LinDesc ld;
        GroupLaunch gl;
        return lin_prepare(*d, ld, gl);
    }
    uint32_t keep, red, plate;
    const int rc = classify(*d, keep, red, plate);
    if (rc == ALAN_OK && d->noise.on) {                  // generated noise (or a carried hand-on): the small kernel or nothing
        SmallDesc sd;
        GroupLaunch gl;
        int mode;
        if (d->noise.on != 2 && d->mode != ALAN_MODE_AFFINE && d->mode != ALAN_MODE_DOT) return ALAN_ERR_BAD_DESC;
        return prepare_small(*d, sd, gl, mode) ? ALAN_OK : ALAN_ERR_UNSUPPORTED;
    }
    return rc;
}

extern "C" size_t alan_reduce_workspace_bytes(const alan_reduce_desc_t *d) {
    if (!d) return 0;
    if (d->mode == ALAN_MODE_BERNOULLI_LINEAR || d->mode == ALAN_MODE_BERNOULLI_LINEAR_GRAD || d->mode == ALAN_MODE_NORMAL_TABLE)
        return 0;
    uint32_t keep, red, plate;
    if (classify(*d, keep, red, plate) != ALAN_OK) return 0;
    {
        alan_reduce_desc_t d2;
        if (split_long_dim(*d, keep, red, plate, d2)) return alan_reduce_workspace_bytes(&d2);
    }
    {
        const int p = peel_dim(*d, keep, red, plate);
        if (p >= 0) {
            alan_tensor_t v;
            int64_t numel;
            peel_layout(*d, keep, p, v, numel);
            return ((size_t)numel * dtype_bytes(v.dtype) + 255) & ~(size_t)255;
        }
    }
    if (!plate || !red) return 0;
    {
        PairDesc pd;
        dim3 grid;
        size_t lds;
        if (pair_prepare(*d, keep, red, plate, pd, grid, lds)) return (pair_workspace_bytes(pd, grid) + 255) & ~(size_t)255;
    }
    {
        Canon c;
        RowsPlan rp;
        if (plan_fused_plate(*d, keep, red, plate, c, rp)) return (rp.partial_bytes + 255) & ~(size_t)255;
    }
    if (d->lse_out.data) return 0;
    int64_t numel = 1;
    for (int i = 0; i < d->ndim; ++i)
        if (d->role[i] != ALAN_REDUCE) numel *= d->size[i];
    return ((size_t)numel * dtype_bytes(d->out.dtype) + 255) & ~(size_t)255;
}

extern "C" int alan_reduce(const alan_reduce_desc_t *d, void *workspace, size_t workspace_bytes, void *stream_) {
    if (!d) return ALAN_ERR_BAD_DESC;
    hipStream_t stream = (hipStream_t)stream_;
    {
        alan_reduce_desc_t d2;
        int64_t n, stride;
        const int p = strip_presum(*d, d2, n, stride);
        if (p == -2) return ALAN_ERR_BAD_DESC;
        if (p >= 0) return run_presum(d2, n, stride, stream, false);
    }
    if (d->mode == ALAN_MODE_NORMAL_TABLE) {
        SmallDesc sd;
        GroupLaunch gl;
        const int rc = table_prepare(*d, sd, gl);
        if (rc != ALAN_OK) return rc;
        const int mode = ALAN_MODE_NORMAL_TABLE;
        return launch_small_multi(&sd, &gl, &mode, 1, stream, nullptr, nullptr, false);
    }
    if (d->mode == ALAN_MODE_BERNOULLI_LINEAR || d->mode == ALAN_MODE_BERNOULLI_LINEAR_GRAD) {
        LinDesc ld;
        GroupLaunch gl;
        const int rc = lin_prepare(*d, ld, gl);
        if (rc != ALAN_OK) return rc;
        EvPair ev;
        ev.start = (hipEvent_t)d->ev_start;
        ev.stop = (hipEvent_t)d->ev_stop;
        if (d->mode == ALAN_MODE_BERNOULLI_LINEAR) {
            // plain terms that meet only in the summed dim, millions of elements: output tiles with the terms staged in LDS
            uint32_t keep = 0, red = 0;
            for (int i = 0; i < d->ndim; ++i) {
                if (d->role[i] == ALAN_KEEP) keep |= 1u << i;
                if (d->role[i] == ALAN_REDUCE) red |= 1u << i;
            }
            PairDesc pd;
            dim3 grid;
            size_t lds;
            if (pair_prepare(*d, keep, red, 0, pd, grid, lds)) return launch_pair(pd, grid, lds, nullptr, 0, stream, ev);
        }
        return d->mode == ALAN_MODE_BERNOULLI_LINEAR ? launch_lin(ld, gl, stream, ev) : launch_lin_grad(ld, stream, ev);
    }
    uint32_t keep, red, plate;
    int rc = classify(*d, keep, red, plate);
    if (rc != ALAN_OK) return rc;
    if (d->noise.on == 2) {                              // a carried hand-on: as a batch of one
        const alan_reduce_desc_t *one = d;
        return alan_reduce_batch(&one, 1, stream_);
    }
    if (d->noise.on) {                                   // generated noise: the small kernel or nothing
        SmallDesc sd;
        GroupLaunch gl;
        int mode;
        if (d->mode != ALAN_MODE_AFFINE && d->mode != ALAN_MODE_DOT) return ALAN_ERR_BAD_DESC;
        if (d->ring_n || !prepare_small(*d, sd, gl, mode)) return ALAN_ERR_UNSUPPORTED;
        sd.noise_on = 1, sd.noise_off = d->noise.offset;
        return launch_small_multi(&sd, &gl, &mode, 1, stream, nullptr, &d->noise, true);
    }
    EvPair ev;
    ev.start = (hipEvent_t)d->ev_start;
    ev.stop = (hipEvent_t)d->ev_stop;

    if (d->ring_n) {
        if (d->ring_n < 0 || !d->ring_slots || !d->ring_counter) return ALAN_ERR_BAD_DESC;
        // one launch writing one value: no plate stage, no workspace (i.e. no two-stage plan), no long-dim split
        alan_reduce_desc_t d2;
        if (plate || alan_reduce_workspace_bytes(d) != 0 || split_long_dim(*d, keep, red, plate, d2))
            return ALAN_ERR_UNSUPPORTED;
        ev.ring_slots = d->ring_slots;
        ev.ring_counter = d->ring_counter;
        ev.ring_n = d->ring_n, ev.ring_and_out = d->ring_and_out;
    }
    if (d->mode == ALAN_MODE_LSE && red == 0) {
        // logsumexp over no dims is the identity (utils.py:217): plain broadcast sum of the factors,
        // followed by the plate sum if any.
        return run_single(*d, keep, plate, ALAN_MODE_SUM, d->out, d->add_const, stream, ev);
    }
    {
        alan_reduce_desc_t d2;
        if (split_long_dim(*d, keep, red, plate, d2)) return alan_reduce(&d2, workspace, workspace_bytes, stream_);
    }
    {
        const int p = peel_dim(*d, keep, red, plate);
        if (p >= 0) {
            alan_tensor_t v;
            int64_t numel;
            peel_layout(*d, keep, p, v, numel);
            if (!workspace || workspace_bytes < (size_t)numel * dtype_bytes(v.dtype)) return ALAN_ERR_WORKSPACE;
            v.data = workspace;
            rc = run_single(*d, keep | (1u << p), red & ~(1u << p), d->mode, v, 0.0, stream);
            if (rc != ALAN_OK) return rc;
            alan_reduce_desc_t s2;
            std::memset(&s2, 0, sizeof(s2));
            s2.mode = d->mode;
            s2.ndim = d->ndim;
            for (int i = 0; i < d->ndim; ++i) {
                const bool k = (keep >> i) & 1;
                s2.size[i] = (k || i == p) ? d->size[i] : 1;
                s2.role[i] = i == p ? ALAN_REDUCE : ALAN_KEEP;
            }
            s2.n_factors = 1;
            s2.factor[0] = v;
            s2.out = d->out;
            return run_single(s2, keep, 1u << p, d->mode, d->out, d->add_const, stream);
        }
    }
    {
        // an output bigger than every factor (two factors meeting only in the reduce dim): tiles staged in LDS
        PairDesc pd;
        dim3 grid;
        size_t lds;
        if (pair_prepare(*d, keep, red, plate, pd, grid, lds))
            return launch_pair(pd, grid, lds, workspace, workspace_bytes, stream, ev);
    }
    if (!plate) return run_single(*d, keep, red, d->mode, d->out, d->add_const, stream, ev);

    // ---- log-sum-exp over REDUCE, then sum over PLATE (logpq.py:128,149)
    {
        // a factor of a few hundred KB is not a stream: the rows kernel's windows and two-stage plate sum cost it 15 us
        // (bus_breakdown's Year plate at K = 100, 20,000 elements) where the lane-group kernel takes 5
        int64_t elems = 1;
        for (int i = 0; i < d->ndim; ++i) elems *= d->size[i];
        SmallPlateDesc sp;
        GroupLaunch gl;
        if (!d->lse_out.data && elems <= (1ll << 17) &&
            small_plate_prepare(*d, sp, gl))
            return launch_small_plate(sp, gl, stream, ev);
    }
    {
        Canon c;
        RowsPlan rp;
        if (plan_fused_plate(*d, keep, red, plate, c, rp))
            return launch_rows(c, rp, ALAN_MODE_LSE, d->add_const, workspace, workspace_bytes, stream, ev);
    }
    {
        SmallPlateDesc sp;
        GroupLaunch gl;
        if (small_plate_prepare(*d, sp, gl)) return launch_small_plate(sp, gl, stream, ev);
    }
    alan_tensor_t v;
    if (d->lse_out.data) {
        v = d->lse_out;
        if (!valid_dtype(v.dtype)) return ALAN_ERR_BAD_DESC;
    } else {
        int64_t numel;
        plate_workspace_layout(*d, v, numel);
        v.dtype = d->out.dtype;
        if (!workspace || workspace_bytes < (size_t)numel * dtype_bytes(v.dtype)) return ALAN_ERR_WORKSPACE;
        v.data = workspace;
    }
    v.scale = 1.f;
    rc = run_single(*d, keep | plate, red, ALAN_MODE_LSE, v, 0.0, stream, ev);
    if (rc != ALAN_OK) return rc;

    alan_reduce_desc_t s2;
    std::memset(&s2, 0, sizeof(s2));
    s2.mode = ALAN_MODE_SUM;
    s2.ndim = d->ndim;
    for (int i = 0; i < d->ndim; ++i) {
        s2.size[i] = d->role[i] == ALAN_REDUCE ? 1 : d->size[i];
        s2.role[i] = d->role[i] == ALAN_PLATE ? ALAN_REDUCE : ALAN_KEEP;
    }
    s2.n_factors = 1;
    s2.factor[0] = v;
    s2.out = d->out;
    return run_single(s2, keep, plate, ALAN_MODE_SUM, d->out, d->add_const, stream);
}

extern "C" int alan_abi_version(void) { return 14; }
extern "C" const char *alan_build_target(void) { return "gfx950"; }
