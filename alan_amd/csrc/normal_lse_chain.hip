// alan_normal_lse_chained: a WHOLE two-level evaluation in one launch --
//
//   prelude   the queued log-prob producers (alan_reduce problems of the small kind: the -(log Q + log K) of the
//             plate's latent, the data likelihood with its linear logits, the parent level's log P - log Q - log K),
//             run by the launch's FIRST workgroups;
//   body      the fused plate step (alan_normal_lse, bf16x3 kernel), whose small factors the prelude writes;
//   tail      the parent's contraction (one or two alan_reduce steps; the first adds the body's per-slice partial sums
//             as it loads them, role ALAN_PRESUM; the last may deliver through a result ring), run by the launch's
//             LAST-ARRIVING workgroup.
//
// What the reference does as lp_getter + reduce_Ks at two plate levels (logpq.py:68-155, reduce_Ks.py:236-298) and
// this library did as three launches (producers / plate step / top level): inside a replayed graph a dependent launch
// costs 4-5 us whatever it computes (tools/replay_floor_probe.py: the plate step alone replays every 14.0 us at K = 30,
// with one trivial kernel in front of it every 18.0 us), so an evaluation of 28-30 us is mostly launches.
//
// In-launch hand-offs (cdna_hip_programming.md G16, the `sc1` forms):
//   prelude -> body   every result store of the prelude is write-through; each of its workgroups drains its stores
//             (s_waitcnt vmcnt(0) in every wave, then the workgroup barrier) and ONE lane adds to state[0] (agent
//             scope).  A body workgroup's thread 0 polls state[0] in front of the barrier that follows its B table --
//             1.5 us into its life, by when the prelude is normally done -- and every small-factor load behind it is an
//             `sc1` load.  Forward progress: the prelude's workgroups hold the launch's LOWEST workgroup ids, so they
//             are dispatched before any workgroup that waits for them and they wait for nobody; the spin is bounded
//             all the same (results NaN and state[2] set past the bound).
//   everyone -> tail  partial sums leave write-through; ONE lane per workgroup adds to state[1] after the storing wave
//             drained; the workgroup whose add returns the last ticket runs an agent-scope acquire and then the tail's
//             steps (a workgroup barrier between two steps: the second reads what this workgroup itself wrote).
//   The last workgroup leaves state[0] = state[1] = 0: the caller zeroes `state` once, never again (no memset node).
#include "normal_lse_chain_impl.h"

namespace alan {

// plan.hip: the single small-kernel launch an alan_reduce problem is, without launching it
int small_problem_prepare(const alan_reduce_desc_t &d, SmallDesc &sd, LinDesc &ld, GroupLaunch &gl, int &mode);

}  // namespace alan

using namespace alan;

namespace {

// ---- prelude problems the body computes in its tiles (normal_lse_x3.h, REC) instead of reading their output ----------
// A problem qualifies when its output IS one of the body's small factors and it is one of the two producers below on
// the body's own value rows.  `f`: which small factor.
int small_index(const alan_normal_lse_desc_t &a, const alan_tensor_t &out, int iM, int iK) {
    for (int f = 0; f < a.n_small; ++f)
        if (a.small[f] == out.data && out.dtype == ALAN_F32 && a.small_sm[f] == (iM >= 0 ? out.stride[iM] : 0) &&
            a.small_sk[f] == (iK >= 0 ? out.stride[iK] : 0))
            return f;
    return -1;
}

// sizes > 1 only: the plate dim (KEEP, size M, value stride v_sm), the K dim (KEEP, size NK, value stride v_sk)
bool find_plate_k(const alan_normal_lse_desc_t &a, const alan_reduce_desc_t &d, const alan_tensor_t &value, int &iM, int &iK,
                  uint32_t &others) {
    iM = iK = -1, others = 0;
    for (int i = 0; i < d.ndim; ++i) {
        if (d.size[i] == 1) continue;
        if (d.role[i] == ALAN_KEEP && iM < 0 && d.size[i] == a.M && value.stride[i] == a.v_sm && a.M > 1)
            iM = i;
        else if (d.role[i] == ALAN_KEEP && iK < 0 && d.size[i] == a.NK && value.stride[i] == a.v_sk && a.NK > 1)
            iK = i;
        else
            others |= 1u << i;
    }
    return (iM >= 0 || a.M == 1) && (iK >= 0 || a.NK == 1);
}

// sum_e log N(value[m,k,e]; loc[m,e], scale[m,e]) scaled and shifted (alan_reduce mode NORMAL / NORMAL_LOGSCALE)
int match_normal_recipe(const alan_normal_lse_desc_t &a, const alan_reduce_desc_t &d, X3Recipes &r) {
    if ((d.mode != ALAN_MODE_NORMAL && d.mode != ALAN_MODE_NORMAL_LOGSCALE) || d.n_factors != 3 || d.ndim > MAXD) return -1;
    if (d.weight.data || d.lse_out.data || d.ring_n) return -1;
    const alan_tensor_t &v = d.factor[0], &l = d.factor[1], &sc = d.factor[2];
    if (v.data != a.value || v.dtype != ALAN_F32 || l.dtype != ALAN_F32 || sc.dtype != ALAN_F32) return -1;
    if (v.scale != 1.f || l.scale != 1.f) return -1;                    // (term weight 1, location not rescaled)
    int iM, iK;
    uint32_t others;
    if (!find_plate_k(a, d, v, iM, iK, others)) return -1;
    int iE = -1;
    for (int i = 0; i < d.ndim; ++i)
        if ((others >> i) & 1) {
            if (iE >= 0 || d.role[i] != ALAN_REDUCE) return -1;
            iE = i;
        }
    if (a.E == 1 ? false : (iE < 0 || d.size[iE] != a.E || v.stride[iE] != a.v_se)) return -1;
    if (a.E == 1 && iE >= 0) return -1;
    if ((iK >= 0 && (l.stride[iK] != 0 || sc.stride[iK] != 0))) return -1;       // (parameters per plate element only)
    const int f = small_index(a, d.out, iM, iK);
    if (f < 0) return -1;
    const int64_t lim = (1ll << 31) - 1;
    const int64_t l_sm = iM >= 0 ? l.stride[iM] : 0, s_sm = iM >= 0 ? sc.stride[iM] : 0;
    const int64_t l_se = iE >= 0 ? l.stride[iE] : 0, s_se = iE >= 0 ? sc.stride[iE] : 0;
    if (l_sm < 0 || s_sm < 0 || l_se < 0 || s_se < 0 || a.M * l_sm + a.E * l_se > lim || a.M * s_sm + a.E * s_se > lim) return -1;
    r.loc = (const float *)l.data, r.scl = (const float *)sc.data;
    r.l_sm = (int32_t)l_sm, r.l_se = (int32_t)l_se, r.s_sm = (int32_t)s_sm, r.s_se = (int32_t)s_se;
    r.has_normal = 1;
    r.n_log_scale = (d.mode == ALAN_MODE_NORMAL_LOGSCALE || sc.scale == 2.f) ? 1 : 0;
    r.n_scale = d.out.scale, r.n_add = (float)d.add_const;
    return f;
}

// sum_n Bernoulli(y[m,n]; logits = sum_e value[m,k,e] x[m,n,e]) scaled and shifted (alan_reduce mode BERNOULLI_LINEAR
// with the one term `value . x`)
int match_linear_recipe(const alan_normal_lse_desc_t &a, const alan_reduce_desc_t &d, X3Recipes &r) {
    if (d.mode != ALAN_MODE_BERNOULLI_LINEAR || d.n_factors != 3 || d.ndim > MAXD) return -1;
    if (d.weight.data || d.lse_out.data || d.ring_n) return -1;
    const alan_tensor_t &y = d.factor[0], &v = d.factor[1], &x = d.factor[2];
    if ((int)v.scale != 1 || (int)x.scale != 1) return -1;             // (both operands of term 1)
    if (v.data != a.value || v.dtype != ALAN_F32 || x.dtype != ALAN_F32 || y.dtype != ALAN_F32) return -1;
    int iM, iK;
    uint32_t others;
    if (!find_plate_k(a, d, v, iM, iK, others)) return -1;
    int iE = -1, iN = -1;
    for (int i = 0; i < d.ndim; ++i)
        if ((others >> i) & 1) {
            if (d.role[i] == ALAN_DOT && iE < 0)
                iE = i;
            else if (d.role[i] == ALAN_REDUCE && iN < 0)
                iN = i;
            else
                return -1;
        }
    if (a.E == 1 ? iE >= 0 : (iE < 0 || d.size[iE] != a.E || v.stride[iE] != a.v_se)) return -1;
    const int64_t N = iN >= 0 ? d.size[iN] : 1;
    if (N > X3_REC_NMAX) return -1;
    if (iN >= 0 && v.stride[iN] != 0) return -1;
    if (iK >= 0 && (x.stride[iK] != 0 || y.stride[iK] != 0)) return -1;
    if (iE >= 0 && y.stride[iE] != 0) return -1;
    const int f = small_index(a, d.out, iM, iK);
    if (f < 0) return -1;
    const int64_t lim = (1ll << 31) - 1;
    const int64_t x_sm = iM >= 0 ? x.stride[iM] : 0, x_sn = iN >= 0 ? x.stride[iN] : 0, x_se = iE >= 0 ? x.stride[iE] : 0;
    const int64_t y_sm = iM >= 0 ? y.stride[iM] : 0, y_sn = iN >= 0 ? y.stride[iN] : 0;
    if (x_sm < 0 || x_sn < 0 || x_se < 0 || y_sm < 0 || y_sn < 0 || a.M * x_sm + N * x_sn + a.E * x_se > lim ||
        a.M * y_sm + N * y_sn > lim)
        return -1;
    r.x = (const float *)x.data, r.y = (const float *)y.data;
    r.x_sm = (int32_t)x_sm, r.x_sn = (int32_t)x_sn, r.x_se = (int32_t)x_se, r.y_sm = (int32_t)y_sm, r.y_sn = (int32_t)y_sn;
    r.N = (int32_t)N;
    r.b_scale = d.out.scale, r.b_add = (float)d.add_const;
    return f;
}

// Does the chained launch take this call?  Fills its kernel argument.
int plan_chain(const alan_normal_lse_desc_t *a, const alan_reduce_desc_t *const *prelude, int32_t n_prelude,
               const alan_reduce_desc_t *const *tail, int32_t n_tail, void *state, ChainPlan &p) {
    if (!a || n_prelude < 0 || n_tail < 0 || (n_prelude && !prelude) || (n_tail && !tail) || !state) return ALAN_ERR_BAD_DESC;
    if (n_prelude > CHAIN_MULTI || n_tail > CHAIN_TAIL) return ALAN_ERR_UNSUPPORTED;
    if (!a->keep_partials || a->counters || a->lse_out || a->ev_start || a->ev_stop) return ALAN_ERR_UNSUPPORTED;
    std::memset(&p.k, 0, sizeof(p.k));
    // ---- prelude problems the body computes in its tiles: their outputs leave the body's list of small factors.  Only
    // where a value tile serves few (loc row, scale tile) units is that cheaper than reading them -- K <= 32, two loc rows
    // per wave: at K = 100 a hundred workgroups would each recompute a plate slice's factors
    alan_normal_lse_desc_t body = *a;
    bool in_tile[CHAIN_MULTI] = {false, false, false, false};
    static const int rec_knob = env_knob("ALAN_CHAIN_REC");                             // ablation knob: 0 = off
    p.rec = false;
    if (rec_knob != 0 && a->NS <= 32 && a->NL >= 8) {
        X3Recipes &r = p.k.rec;
        bool gone[4] = {false, false, false, false};
        for (int i = 0; i < n_prelude; ++i) {
            if (!prelude[i]) return ALAN_ERR_BAD_DESC;
            X3Recipes trial = r;
            int f = -1;
            if (!r.has_normal) f = match_normal_recipe(*a, *prelude[i], trial);
            if (f < 0 && r.N == 0) trial = r, f = match_linear_recipe(*a, *prelude[i], trial);
            if (f < 0 || gone[f]) continue;
            if (2 * (trial.has_normal ? a->E : 0) + trial.N * a->E + trial.N > 64 * X3_REC_SLOTS) continue;
            // (no OTHER prelude problem may write the same tensor, nor any read it: nothing does -- producers read
            // samples, parameters and data only, and the caller hands each output to one consumer)
            r = trial, in_tile[i] = true, gone[f] = true, p.rec = true;
        }
        if (p.rec) {
            int n = 0;
            for (int f = 0; f < a->n_small; ++f)
                if (!gone[f]) body.small[n] = a->small[f], body.small_sm[n] = a->small_sm[f], body.small_sk[n] = a->small_sk[f], ++n;
            body.n_small = n;
        }
    }
    int rc = nl_x3_prepare(body, a->out, p.xp);
    if (rc != ALAN_OK) return rc;
    if (p.rec && !(p.xp.nlw == 2 && !p.xp.flat)) return ALAN_ERR_UNSUPPORTED;         // (the shapes REC is instantiated for)
    if (p.rec) p.xp.lds += 4 * sizeof(float) * (p.xp.eq == 4 ? X3RecLayout<4>::FLOATS : p.xp.eq == 8 ? X3RecLayout<8>::FLOATS :
                                                  p.xp.eq == 10 ? X3RecLayout<10>::FLOATS : p.xp.eq == 12 ? X3RecLayout<12>::FLOATS :
                                                  X3RecLayout<17>::FLOATS);
    {
        // diagnostic knob (tools/chain_parts.py): 1 = the in-tile producers' kernel with nothing to compute, 2 = only the
        // Normal term, 3 = only the linear-logits term -- wrong results, for timing the parts
        static const int nop_knob = env_knob("ALAN_CHAIN_REC_NOP");
        if (p.rec && nop_knob != ENV_UNSET) {
            if (nop_knob == 1 || nop_knob == 3) p.k.rec.has_normal = 0;
            if (nop_knob == 1 || nop_knob == 2) p.k.rec.N = 0;
        }
    }
    p.k.d = p.xp.x;
    ChainArgs &c = p.k.c;
    c.gx = p.xp.gx, c.gy = p.xp.gy, c.n_main = p.xp.gx * p.xp.gy * p.xp.gz;
    c.gxd = make_fastdiv(c.gx), c.gyd = make_fastdiv(c.gy);
    c.state = (int32_t *)state;
    // ---- the prelude: every problem one small-kernel launch, at most one of them the linear-logits producer
    SmallDesc sd[CHAIN_MULTI];
    GroupLaunch gl[CHAIN_MULTI];
    LinDesc lin;
    int mode[CHAIN_MULTI];
    bool have_lin = false;
    std::memset(sd, 0, sizeof(sd));
    int na = 0;
    for (int i = 0; i < n_prelude; ++i) {
        if (!prelude[i]) return ALAN_ERR_BAD_DESC;
        if (prelude[i]->ring_n || prelude[i]->ev_start || prelude[i]->ev_stop) return ALAN_ERR_UNSUPPORTED;
        if (in_tile[i]) continue;
        LinDesc l1;
        rc = small_problem_prepare(*prelude[i], sd[na], l1, gl[na], mode[na]);
        if (rc != ALAN_OK) return rc;
        if (mode[na] == ALAN_MODE_BERNOULLI_LINEAR) {
            if (have_lin) return ALAN_ERR_UNSUPPORTED;
            have_lin = true, lin = l1;
        }
        for (int f = 0; f < body.n_small; ++f)
            if (body.small[f] == prelude[i]->out.data) c.body_waits = 1;
        ++na;
    }
    c.pre_blocks = na ? fill_small_multi(p.k.pre, sd, gl, mode, na, have_lin ? &lin : nullptr) : 0;
    // A workgroup of the prelude per workgroup of its problems (they are chains of load latencies: only side by side are
    // they quick), as long as that stays a fraction of the launch; each holds one of the chip's 512 slots for its
    // lifetime, which is how much later the body's last workgroups start.
    static const int aux_knob = env_knob("ALAN_CHAIN_AUX");                            // tuning knob: prelude workgroups
    const uint32_t aux_max = aux_knob != ENV_UNSET ? (uint32_t)std::max(1, aux_knob) : 2048u;
    if (c.pre_blocks > 8 * aux_max) return ALAN_ERR_UNSUPPORTED;
    c.n_aux = (int32_t)std::min<uint32_t>(c.pre_blocks, aux_max);
    // ---- the tail: a chain of small problems, each one launch of one workgroup's worth of virtual workgroups
    c.n_tail = n_tail;
    for (int s = 0; s < n_tail; ++s) {
        if (!tail[s]) return ALAN_ERR_BAD_DESC;
        if (tail[s]->ev_start || tail[s]->ev_stop) return ALAN_ERR_UNSUPPORTED;
        LinDesc l1;
        GroupLaunch g1;
        int m1;
        rc = small_problem_prepare(*tail[s], p.k.tail[s], l1, g1, m1);
        if (rc != ALAN_OK) return rc;
        if (m1 != ALAN_MODE_LSE && m1 != ALAN_MODE_SUM) return ALAN_ERR_UNSUPPORTED;
        if (g1.grid > 64) return ALAN_ERR_UNSUPPORTED;      // (one workgroup walks them: a long tail belongs in a launch)
        if (tail[s]->ring_n) {
            if (tail[s]->ring_n < 0 || !tail[s]->ring_slots || !tail[s]->ring_counter) return ALAN_ERR_BAD_DESC;
            if (p.k.tail[s].n_out != 1 || s != n_tail - 1) return ALAN_ERR_UNSUPPORTED;
            p.k.tail[s].ring_slots = (float *const *)tail[s]->ring_slots;
            p.k.tail[s].ring_counter = (int32_t *)tail[s]->ring_counter;
            p.k.tail[s].ring_n = tail[s]->ring_n;
        }
        c.tail_mode[s] = m1, c.tail_logG[s] = g1.logG, c.tail_block[s] = g1.block ? 1 : 0, c.tail_blocks[s] = g1.grid;
    }
    p.grid = (uint32_t)c.n_aux + c.n_main;
    p.syncfree = n_tail == 0 && !c.body_waits && p.rec;
    return ALAN_OK;
}

}  // namespace

extern "C" int alan_normal_lse_chained_check(const alan_normal_lse_desc_t *a, const alan_reduce_desc_t *const *prelude,
                                             int32_t n_prelude, const alan_reduce_desc_t *const *tail, int32_t n_tail) {
    ChainPlan p;
    int dummy[4];
    const int rc = plan_chain(a, prelude, n_prelude, tail, n_tail, dummy, p);
    return rc != ALAN_OK ? rc : p.syncfree ? ALAN_OK : ALAN_CHAIN_HANDOFFS;
}

extern "C" int alan_normal_lse_chained(const alan_normal_lse_desc_t *a, const alan_reduce_desc_t *const *prelude,
                                       int32_t n_prelude, const alan_reduce_desc_t *const *tail, int32_t n_tail, void *state,
                                       void *stream_) {
    ChainPlan p;
    int rc = plan_chain(a, prelude, n_prelude, tail, n_tail, state, p);
    if (rc != ALAN_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    switch (p.xp.eq) {
        case 4: return chain_launch_eq4(p, stream);
        case 8: return chain_launch_eq8(p, stream);
        case 10: return chain_launch_eq10(p, stream);
        case 12: return chain_launch_eq12(p, stream);
        default: return chain_launch_eq17(p, stream);
    }
}
