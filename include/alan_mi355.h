/*
 * alan_mi355.h -- C ABI of libalan_mi355.so: the MI355X (gfx950) implementation of alan's
 * tensorised marginal-likelihood hot path.
 *
 * The reference (alan-ppl/alan) is pure Python and has no FFI of its own; these entry points
 * are what a binding for the path would call, one per reference function on the path:
 *
 *   alan_reduce(mode = ALAN_MODE_LSE)      reduce_Ks.py:249-251  logsumexp_sum(Ks, *lps)
 *                                          utils.py:207-222       logsumexp_dims  (eps-in-log epilogue)
 *                                          utils.py:224-225       logmeanexp_dims (add_const = -sum log K)
 *                                          Sampler.py:118-134     SamplerMP.reduce_logQ
 *     ... with dims of role ALAN_PLATE     logpq.py:149           lp.sum(new_platedim) fused behind the LSE
 *   alan_reduce(mode = ALAN_MODE_SUM)      logpq.py:149-153       plate sum / Split accumulate (prev_lpq + lp)
 *   alan_reduce(mode = ALAN_MODE_WEXPSUM)  backward of the above (what autograd derives from utils.py:218-220):
 *                                          grad_f = sum_{dims not in f} grad_out * exp(sum_f lp_f - lse)
 *   alan_reduce(mode = ALAN_MODE_NORMAL)   TorchDimDist.py:127-162 log_prob of a Normal over the K cross-product with
 *                                          the event-dim sum (utils.py:147-152) fused in: the factor PRODUCER
 *   alan_chain_logmmexp_batched            utils.py:478-510 chain_logmmexp  (+ logpq.py:139 logsumexp(-1))
 *   alan_chain_logmmexp_backward_batched   autograd through the same
 *
 * Conventions
 *   - Plain pointers and sizes only; every pointer is DEVICE memory owned by the caller.
 *     The library never allocates, frees or synchronises; all work is enqueued on `stream`.
 *   - Tensors are described over one shared list of dims ("the space"): size[d], role[d] and, per
 *     tensor, an element stride per dim (0 = the tensor does not carry that dim => broadcast).
 *   - Return value: 0 on success, negative alan_status_t otherwise.  Nothing throws across the ABI.
 *   - Re-entrant and thread-safe: no global mutable state; the device is whatever is current.
 */
#ifndef ALAN_MI355_H
#define ALAN_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALAN_MAX_DIMS 8
#define ALAN_MAX_FACTORS 6

typedef enum {
    ALAN_OK = 0,
    ALAN_ERR_BAD_DESC = -1,       /* malformed descriptor (ndim, sizes, roles, null pointers) */
    ALAN_ERR_UNSUPPORTED = -2,    /* dtype / size outside what the kernels implement */
    ALAN_ERR_WORKSPACE = -3,      /* caller-provided workspace too small */
    ALAN_ERR_LAUNCH = -4          /* hipLaunchKernel reported an error */
} alan_status_t;

typedef enum { ALAN_F32 = 0, ALAN_F64 = 1 } alan_dtype_t;

/* role of a dim of the space */
typedef enum {
    ALAN_KEEP = 0,    /* survives into the output */
    ALAN_REDUCE = 1,  /* reduced by the mode's operator (log-sum-exp / sum / weighted exp-sum) */
    ALAN_PLATE = 2,   /* ALAN_MODE_LSE only: summed AFTER the log-sum-exp (logpq.py:149) */
    ALAN_DOT = 3,     /* ALAN_MODE_BERNOULLI_LINEAR only: an event dim the logits are contracted over */
    ALAN_PRESUM = 4   /* ALAN_MODE_LSE / ALAN_MODE_SUM only: a dim carried by exactly ONE factor (zero stride in every
                         other tensor), summed BEFORE that factor enters the sum of factors -- the per-slice partial
                         results alan_normal_lse leaves with keep_partials (logpq.py:149: the plate sum, finished by the
                         launch that consumes it instead of by a launch of its own).  At most one such dim; no
                         ALAN_PLATE dim beside it.  ALAN_ERR_UNSUPPORTED (alan_reduce_check says so beforehand) unless
                         the problem takes the small single-launch kernel: the caller then sums the factor first */
} alan_role_t;

typedef enum {
    ALAN_MODE_LSE = 0,     /* out = log(sum_R exp(x - max_R x) + eps(dtype)) + max_R x,  x = sum_f scale_f * factor_f */
    ALAN_MODE_SUM = 1,     /* out = sum_R x */
    ALAN_MODE_WEXPSUM = 2, /* out = sum_R weight * exp(x) */
    ALAN_MODE_NORMAL = 3,  /* fused factor PRODUCER (TorchDimDist.py:127-162 + utils.py:147-152 for td.Normal):
                              3 factors (value, loc, scale), or 6 = two such terms (log P and log Q of one variable);
                              out = sum_t value_t.scale * sum_R [ -(value-loc)^2 / (2 scale^2) - log(scale) - log(sqrt(2 pi)) ],
                              R = the event/batch dims; the [..., K, K, K, d] broadcast is never materialised.
                              A scale factor whose own .scale field is 2 holds log(scale) (see NORMAL_LOGSCALE); a loc
                              factor's .scale field c multiplies it (loc = c * stored value: a lambda `c * prev`). */
    ALAN_MODE_BERNOULLI = 4, /* fused factor PRODUCER for td.Bernoulli(logits=...) (same reference lines):
                              exactly 2 factors (value, logits);
                              out = sum_R [ logsigmoid(logits) - (1 - value) * logits ]
                              (= -binary_cross_entropy_with_logits, what torch's Bernoulli.log_prob evaluates);
                              R = event/batch dims and, for a data-only plate, the plate dims (logpq.py:149) */
    ALAN_MODE_NORMAL_LOGSCALE = 5, /* ALAN_MODE_NORMAL whose third factor is log(scale): the exp() transform of a
                              learned scale parameter (Param.py:18-25, transformation=t.exp) folded into the producer */
    ALAN_MODE_PRODUCER_GRAD = 6, /* BACKWARD of a producer with respect to ONE of its arguments (what autograd derives
                              from TorchDimDist.py:127-162): factors = (G, value, loc, scale) or (G, value, logits), G
                              the upstream gradient laid out like the producer's output;  out = out.scale * sum_R
                              G * d log-prob / d argument, KEEP = that argument's dims.  Which one: factor[0].scale =
                              1: Normal d/d value, 2: d/d loc, 3: d/d scale (d/d log scale when factor[3].scale == 2),
                              4: Bernoulli d/d logits */
    ALAN_MODE_BERNOULLI_LINEAR = 7, /* ALAN_MODE_BERNOULLI whose logits are a sum of terms computed in the launch instead
                              of by the model's lambda beforehand (movielens `lambda z, x: z @ x`, bus_breakdown
                              `alpha + phi @ bus_company_name + psi @ run_type`: TorchDimDist.py:127-162 evaluating
                              the lambda through torchdim, then Bernoulli.log_prob).  factor[0] = value; the others
                              are term operands, factor[i].scale = 1-based term number: a term with one operand is
                              that operand (no ALAN_DOT strides); a term with two operands, adjacent in the list,
                              is sum over ONE ALAN_DOT dim of their product.  At most 3 terms; fp32 only
                              (ALAN_ERR_UNSUPPORTED otherwise: evaluate the logits and use ALAN_MODE_BERNOULLI).
                              out = out.scale * sum_R [ logsigmoid(l) - (1 - value) * l ] + add_const */
    ALAN_MODE_DOT = 8,     /* out = sum_R [ (factor_0 [+ factor_4]) * factor_1 [* g(factor_2)] [+ factor_3.scale * factor_3] ]
                              (+ add_const): 2 factors, or 3 with g as in ALAN_MODE_AFFINE (the gradient of a reparameterised draw with
                              respect to its log-scale: sum G * eps * exp(raw)), or 4 with one more SUMMAND (that
                              gradient's other contribution -- the -1 per unit of log q's upstream gradient that
                              TorchDimDist.py:127-162 gives the log-scale of a variable's own draw -- so that one launch
                              writes the parameter's whole gradient), or 5: the upstream gradient given as the sum of two
                              tensors (a sample used by two consumers: factor_3.scale = 0 when there is no summand).
                              A term of such logits whose
                              operands lack some dim of the likelihood's index space, evaluated once (what the lambda's
                              `phi @ bus_company_name` is); small ones join alan_reduce_batch launches */
    ALAN_MODE_AFFINE = 10, /* out = sum_R [ factor_0 + factor_1 * g(factor_2) ] (+ add_const), g = exp where factor_2.scale == 2
                              (its tensor holds a log-scale), else the identity: the reparameterised draw
                              x = loc + eps * scale of a td.Normal (TorchDimDist.py:88-125 d.rsample; Param.py:18-25 the
                              exp transformation of an OptParam scale) written straight in the sample's layout; several
                              variables' draws join one alan_reduce_batch launch.  Exactly 3 factors. */
    ALAN_MODE_NORMAL_TABLE = 11, /* not a reduction: builds the scale table of a fused plate step (alan_normal_lse_desc_t.
                              scale_table) from its scale argument -- one KEEP dim (the scale rows, <= 32) and one REDUCE
                              dim (the event, <= 32), ONE fp32 factor = scale, or log(scale) where factor_0.scale == 2
                              (as g in ALAN_MODE_AFFINE); out.data = the table (alan_normal_lse_table_bytes bytes, 16-byte
                              aligned, layout the library's own; out's strides are not read).  A problem like any other
                              small one for alan_reduce_batch: it rides in the launch of the plate step's producers. */
    ALAN_MODE_BERNOULLI_LINEAR_GRAD = 9  /* backward of ALAN_MODE_BERNOULLI_LINEAR with respect to the FIRST operand `a`
                              of its FIRST dot term (movielens: z of `z @ x`) -- what autograd derives from the lambda's
                              batched matmul and TorchDimDist.py:127-162.  The forward's factors and roles unchanged;
                              weight = the upstream gradient over the KEEP dims; out = the gradient, laid out over the
                              KEEP dims and the DOT dim like `a` (out.scale = the forward's out.scale):
                                  out[keep, e] = out.scale * sum_R weight[keep] * (value - sigmoid(l)) * b[..., e]
                              `a` must carry every KEEP dim and no REDUCE dim, the dot at most 32 events
                              (ALAN_ERR_UNSUPPORTED otherwise: the caller differentiates the lambda itself) */
} alan_mode_t;
/* Producer modes (NORMAL, NORMAL_LOGSCALE, BERNOULLI) write  out = out.scale * sum_R(log-prob) + add_const, so the
 * "-(log Q + log K)" of logpq.py:234-235 costs no extra pass; out.scale must be 1 in the other modes. */

typedef struct {
    const void *data;                 /* device pointer to element 0 */
    int32_t dtype;                    /* alan_dtype_t */
    float scale;                      /* factor enters the sum as scale * value (1 or -1) */
    int64_t stride[ALAN_MAX_DIMS];    /* element stride per dim of the space; 0 = broadcast */
} alan_tensor_t;

/* Standard-normal noise generated inside the launch instead of read from memory (ALAN_MODE_AFFINE / ALAN_MODE_DOT only:
 * the draws x = loc + eps * scale of Problem.sample, TorchDimDist.py:88-125, and the reparameterised gradient's
 * sum G * eps * scale).  With on != 0, factor 1 of the problem is VIRTUAL: its strides place every element at an offset
 * as usual, nothing is loaded, and the element at offset o is
 *     eps(o) = Box-Muller(Philox4x32-10(key = seed, counter = (i >> 2, 0x414c414e, 0)))[i & 3],  i = offset + counter + o
 * -- a pure function of (seed, i): the backward of a draw regenerates the noise the forward used from the same numbers.
 * factor[1].data must still point at valid device memory (it is not read through).
 *   cell     optional device uint64[2] = {counter, seed}: generator state that lives on the device, so that a launch
 *            replayed from a HIP graph draws fresh noise every replay (and can be re-seeded between replays).  Given:
 *            counter and seed are read from it and the `seed` field is ignored.  NULL: counter = 0, seed = `seed`.
 *            The launch only reads it;
 *   receipt  optional device uint64[2]: receives the counter and the seed this launch used (what a later launch of the
 *            same replay -- the backward -- passes as its cell);
 *   advance  optional device uint64 holding the ADDRESS of a uint64[2]: that slot receives {counter + advance_by, seed}
 *            -- the cell of the NEXT launch that draws (of the first one of the next replay, for the last).  It must not
 *            be this launch's own cell: workgroups of the launch may still be reading that (which is why the state is
 *            handed on from slot to slot instead of advanced in place -- nothing waits for anything).  The address is
 *            read on the device, so the caller can close the ring of slots after a capture has ended.  In
 *            alan_reduce_batch every problem with noise must name the same seed / cell / receipt / advance; the batch's
 *            last launch that holds such a problem does the handing on.
 *   on = 2   on a problem of ANY mode, beside no problem that draws: the problem itself is computed as usual and the
 *            launch that carries it also copies {counter, seed} from `cell` to the slot `advance` names (seed, offset,
 *            receipt, advance_by ignored) -- alan_noise_handon without a launch of its own.
 * ALAN_ERR_UNSUPPORTED (nothing enqueued; alan_reduce_check says so beforehand) when the problem does not take the
 * small single-launch kernel: the caller then draws the noise itself. */
/* (alan_noise_handon, declared below: copies one slot {counter, seed} to another -- for a replayed sequence that holds a
 * SINGLE launch with generated noise, whose `advance` slot must not be its own cell: the launch hands on to a second
 * slot and this one-thread launch, anywhere behind it, copies that back.) */
typedef struct {
    int32_t on;
    uint32_t advance_by;
    uint64_t seed, offset;
    const void *cell;
    void *receipt, *advance;
} alan_noise_t;

typedef struct {
    int32_t mode;                         /* alan_mode_t */
    int32_t ndim;                         /* <= ALAN_MAX_DIMS */
    int64_t size[ALAN_MAX_DIMS];
    int32_t role[ALAN_MAX_DIMS];          /* alan_role_t */
    int32_t n_factors;                    /* 1 .. ALAN_MAX_FACTORS */
    alan_tensor_t factor[ALAN_MAX_FACTORS];
    alan_tensor_t weight;                 /* ALAN_MODE_WEXPSUM: multiplicative weight; data NULL otherwise */
    alan_tensor_t out;                    /* written; strides over KEEP dims; dtype = compute dtype */
    alan_tensor_t lse_out;                /* optional (data may be NULL). With PLATE dims: receives the per-(KEEP,PLATE)
                                             log-sum-exp values (the backward's saved tensor); strides over KEEP+PLATE */
    double add_const;                     /* added to every output element */
    void *ev_start, *ev_stop;             /* optional hipEvent_t pair (NULL = off): recorded on `stream` immediately
                                             before / after the DOMINANT kernel of this call (the one that streams the
                                             largest factor), so a caller can time that kernel alone (bench.py) */
    /* Optional result ring (ring_n = 0: off), for a call that is REPLAYED from a HIP graph and produces ONE fp32
     * value.  ring_slots: device array of ring_n float*; ring_counter: device int32 in [0, ring_n).  The launch writes
     * its value through ring_slots[*ring_counter] instead of `out` and then advances the counter (mod ring_n), so
     * consecutive replays of one graph deliver their results to different addresses and the caller need not copy a
     * result out before the next replay.  ALAN_ERR_UNSUPPORTED (nothing enqueued) unless the call is one
     * single-workgroup launch. */
    void *ring_slots, *ring_counter;
    int32_t ring_n;
    int32_t ring_and_out;                 /* != 0: the value goes through the ring AND to `out` (a caller that keeps `out` for
                                             a backward -- the ELBO of a training iteration -- and hands the ring's slot on) */
    alan_noise_t noise;                   /* on = 0: off */
} alan_reduce_desc_t;

/* Bytes of scratch alan_reduce() needs for this descriptor (0 is possible). */
size_t alan_reduce_workspace_bytes(const alan_reduce_desc_t *desc);

/* What alan_reduce() would return for this descriptor's shape, without enqueuing anything (a caller that queues
 * launches for alan_reduce_batch asks first whether a ALAN_MODE_BERNOULLI_LINEAR problem is taken). */
int alan_reduce_check(const alan_reduce_desc_t *desc);

/* Enqueue the reduction.  `workspace` must be at least alan_reduce_workspace_bytes(desc) bytes,
 * 256-byte aligned, and stay alive until the stream has passed this call. */
int alan_reduce(const alan_reduce_desc_t *desc, void *workspace, size_t workspace_bytes, void *stream);

/* n <= 64 INDEPENDENT alan_reduce problems (no one reads another's output; none needs a workspace or carries timing
 * events).  The small single-stage ones among them -- the per-variable log-prob producers of a plate
 * (TorchDimDist.py:127-162 per variable, logpq.py:221-222), each a launch-latency-bound kernel of its own otherwise --
 * go out first, up to 8 problems per kernel launch; the others follow as alan_reduce would launch them. */
int alan_reduce_batch(const alan_reduce_desc_t *const *descs, int32_t n, void *stream);

/* Backward of an ALAN_MODE_LSE call (with or without PLATE dims) with respect to EVERY factor in one pass over the
 * largest one -- what autograd derives from utils.py:218-220 + logpq.py:149:
 *     grad factor_f = sum_{dims not in f}  weight * exp(sum_f factor_f - lse)
 *   fwd          the forward descriptor; fwd.weight = upstream gradient (strides over the KEEP dims), fwd.lse_out =
 *                the per-(KEEP,PLATE) log-sum-exp values the forward saved (an INPUT here); fwd.out is not used
 *   grad[f]      where factor f's gradient goes (data NULL = not wanted); same strides as factor f
 * Returns ALAN_ERR_UNSUPPORTED when the problem does not have the shape of the streaming kernel (one contiguous
 * REDUCE dim in the largest factor, every other factor constant over the KEEP dims, fp32); the caller then issues
 * one ALAN_MODE_WEXPSUM call per factor. */
typedef struct {
    alan_reduce_desc_t fwd;
    alan_tensor_t grad[ALAN_MAX_FACTORS];
} alan_backward_desc_t;
size_t alan_reduce_backward_workspace_bytes(const alan_backward_desc_t *desc);
int alan_reduce_backward(const alan_backward_desc_t *desc, void *workspace, size_t workspace_bytes, void *stream);

/* The hierarchical-Normal plate step with the factor producer fused in (never materialises the [plate, K, K, K]
 * log-prob factor): replaces TorchDimDist.py:127-162 + utils.py:147-152 (the Normal log-prob over the K cross
 * product), reduce_Ks.py:249-251 + utils.py:218-220 (log-sum-exp over the child K) and logpq.py:149 (plate sum) for
 *     out[l, s] = sum_m LSE_k( log N(value[m,k,:]; loc[l,:], scale[s,:]) + sum_f small_f[m,k] ) + add_const
 * All tensors fp32; strides in elements; small factors may have stride 0 along m or k (a caller holding an fp64
 * small factor -- the likelihood of fp64 observations -- converts it first: the kernel's arithmetic is fp32 either way).
 * scale holds log(scale) when log_scale != 0.  lse_out (optional) receives the per-plate-element values
 * LSE_k(...)[m, l, s], contiguous [M, NL, NS] fp32: what alan_normal_lse_backward needs.  ALAN_ERR_UNSUPPORTED (event
 * length > 32, ...) -> produce the factor with ALAN_MODE_NORMAL and call alan_reduce instead. */
typedef struct {
    const void *value;  int64_t v_sm, v_sk, v_se;      /* [M, NK, E] */
    const void *loc;    int64_t l_sl, l_se;            /* [NL, E]    */
    const void *scale;  int64_t s_ss, s_se;            /* [NS, E]    */
    int32_t log_scale, n_small;                        /* n_small <= 4 */
    const void *small[4]; int64_t small_sm[4], small_sk[4];   /* [M, NK] each */
    int64_t M, NK, NL, NS, E;
    void *out;          int64_t o_sl, o_ss;            /* [NL, NS]   */
    void *lse_out;                                     /* optional [M, NL, NS] fp32 contiguous */
    double add_const;
    void *ev_start, *ev_stop;                          /* optional hipEvent_t pair (NULL = off) recorded immediately
                                                          before / after the MFMA kernel of the call (forward, or the
                                                          backward when the descriptor sits in a backward desc) */
    int32_t keep_partials;                             /* != 0: no second launch -- out receives the launch's partial
                                                          sums, [alan_normal_lse_n_partials(desc), NL, NS] fp32
                                                          contiguous (o_sl, o_ss, add_const not used), for a consumer
                                                          that adds them itself (alan_reduce, role ALAN_PRESUM) */
    const void *scale_table;                           /* optional: what an ALAN_MODE_NORMAL_TABLE problem built from THIS
                                                          scale / log_scale earlier on the stream (the waves then load
                                                          their matrix operand instead of building it behind a barrier:
                                                          about 1 us of a 9.5 us launch at K = 30); same bits either way;
                                                          ignored where alan_normal_lse_table_bytes is 0, and by the
                                                          backward */
} alan_normal_lse_desc_t;
size_t alan_normal_lse_workspace_bytes(const alan_normal_lse_desc_t *desc);
/* Size of the scale table this call can use (0: it cannot -- more than 32 scale rows, or a shape the library declines). */
size_t alan_normal_lse_table_bytes(const alan_normal_lse_desc_t *desc);
/* How many partial results per output a keep_partials call leaves (0: the library declines the shape). */
int64_t alan_normal_lse_n_partials(const alan_normal_lse_desc_t *desc);
int alan_normal_lse(const alan_normal_lse_desc_t *desc, void *workspace, size_t workspace_bytes, void *stream);

/* Backward of alan_normal_lse with respect to EVERY input, in one pass that recomputes the log-prob tiles on the matrix
 * cores as the forward does and never writes the [M, NL, NS, NK] factor or its gradient -- what autograd derives from
 * TorchDimDist.py:127-162 (torch.distributions.Normal.log_prob) + utils.py:147-152 + reduce_Ks.py:249-251 +
 * utils.py:218-220 + logpq.py:149.  With  X[m,l,s,k] = grad_out[l,s] * exp(log N(...) + sum_f small_f[m,k] - lse[m,l,s]):
 *     grad_small[m,k]   = sum_{l,s}   X                      (gradient wrt the SUM of the small factors)
 *     grad_value[m,k,e] = sum_{l,s}   X * -(value - loc) / scale^2
 *     grad_loc[l,e]     = sum_{m,k,s} X *  (value - loc) / scale^2
 *     grad_scale[s,e]   = sum_{m,k,l} X * ((value - loc)^2 / scale^3 - 1 / scale)      (times scale when fwd.log_scale:
 *                                                                                        the gradient wrt log(scale))
 *   fwd         the forward's descriptor (out, o_sl, o_ss, lse_out, add_const are not used)
 *   lse         [M, NL, NS] fp32 contiguous: the forward's lse_out
 *   grad_out    [NL, NS] fp32 with element strides (g_sl, g_ss)
 *   grad_*      fp32, contiguous, any of them NULL = not wanted
 * Deterministic (per-workgroup partials + a second stage, no float atomics). */
typedef struct {
    alan_normal_lse_desc_t fwd;
    const void *lse;
    const void *grad_out; int64_t g_sl, g_ss;
    void *grad_value;       /* [M, NK, E] */
    void *grad_loc;         /* [NL, E]    */
    void *grad_scale;       /* [NS, E]    */
    void *grad_small;       /* [M, NK]    */
} alan_normal_lse_backward_desc_t;
size_t alan_normal_lse_backward_workspace_bytes(const alan_normal_lse_backward_desc_t *desc);
int alan_normal_lse_backward(const alan_normal_lse_backward_desc_t *desc, void *workspace, size_t workspace_bytes,
                             void *stream);

/* Timeseries plate: utils.py:478-510 chain_logmmexp + the t.logsumexp(., -1) of logpq.py:139, for a BATCH of B
 * independent chains (B = 1: the reference's call; B > 1: a timeseries plate nested under other plates or carrying parent
 * K dims -- logpq.py:133-135: lp.order(T, K_init, K_curr) leaves every other torchdim as a batch dim of the matmuls in
 * utils.py:503-507).
 *   ms         [B, T, K, K] with element strides (sB, sT, sRow, sCol); ms[b][t][i][j] = log weight of going from
 *              particle i of step t-1 to particle j of step t
 *   out_chain  optional [B, K, K] contiguous: the log of the ordered matrix product  (chain_logmmexp)
 *   out_vec    optional [B, K]   contiguous: logsumexp(chain, -1)                    (what the ELBO uses)
 * The reference's own pairwise tree (an odd leftover carried to the end, utils.py:488-495) with its normalisation and
 * eps-in-log (utils.py:503-507): where that eps floors entries the bracketing matters, so it is kept.  The workspace
 * receives EVERY round of the tree (alan_chain_batched_workspace_bytes is about the size of ms); keep it if a backward
 * is to follow.  B <= 65535.
 *
 * alan_chain_logmmexp_backward_batched: the gradient with respect to ms -- what autograd derives from utils.py:478-510
 * (+ logpq.py:139), INCLUDING the paths through the eps floor and through amax.  `tree` is the forward's workspace,
 * untouched since.  The upstream gradient is grad_vec [B, K] (of out_vec; then out_vec must be given) and / or grad_chain
 * [B, K, K] (of out_chain); both given = their sum.  grad_ms receives [B, T, K, K] contiguous.
 * fp32, 2 <= rounds: TWO launches whatever T is -- one that marks the workspace as not yet written, one with a workgroup per
 * node of the tree (ABI 14; a launch per round before: 122 -> 72 us at T = 1000, K = 30).  A node waits for its parent's
 * gradient in memory for a BOUNDED time (0.2 s: it then yields NaN, as does everything below it -- never a hang); the
 * launch relies on workgroups starting in index order, not on all of them being resident. */
size_t alan_chain_batched_workspace_bytes(int64_t B, int64_t T, int64_t K, int32_t dtype);
int alan_chain_logmmexp_batched(const void *ms, int32_t dtype, int64_t B, int64_t T, int64_t K,
                                int64_t sB, int64_t sT, int64_t sRow, int64_t sCol,
                                void *out_chain, void *out_vec,
                                void *workspace, size_t workspace_bytes, void *stream);
/* The forward with the chain's input given as the SUM of up to 3 terms -- the factors of a timeseries plate, which
 * the reference adds into one [T, K_init, K] tensor first (reduce_Ks with no K to sum, logpq.py:128): the first round
 * adds them on load.  terms[i] is [B, T, K, K] with element strides strides[4 i .. 4 i + 3] = (sB, sT, sRow, sCol), 0
 * where a term lacks a dim.  Workspace as alan_chain_batched_workspace_bytes.  (No backward of its own: with
 * gradients to record, add the factors first.)
 * `normal` (optional): one MORE term that is computed on load instead of read: the transition log-prob of a timeseries
 * `ts ~ Normal(c * prev, scale)` (Timeseries.py:205-245 evaluating TorchDimDist.py:127-162 on the [T, K_init, K] cross
 * product): log N(value; loc_mul * loc, scale), each operand a [B, T, K_init, K] view given by four element strides
 * (0 where it lacks a dim: value = x[t, k] has no K_init stride, loc = prev[t, k_init] no K stride).  That factor --
 * 40 MB at T=1000, K=100 -- is then never written.  With loc0 the location is the previous state without its
 * concatenation: loc0 at step 0, loc shifted by one step after it. */
typedef struct {
    const void *value, *loc, *scale;      /* dtype as the terms */
    int64_t v_stride[4], l_stride[4], s_stride[4];     /* (sB, sT, sRow, sCol) */
    double loc_mul;
    int32_t log_scale;                    /* `scale` holds log(scale) */
    const void *loc0;                     /* optional: the location of step 0, a [B, K_init, K] view (l0_stride[1] unused);
                                             steps t >= 1 then read `loc` at step t - 1.  What Timeseries.py:205-245 builds
                                             as cat(init, x[:-1]) -- the previous state -- taken from its two sources */
    int64_t l0_stride[4];
} alan_chain_normal_t;
/* `fin` (optional): the PARENT's contraction of the chain's result run by the chain's last launch (one workgroup) behind
 * its last round -- the evaluation's final reduce_Ks when the timeseries plate sits under the top level (Sample.py:69-86,
 * reduce_Ks.py:249-251 + utils.py:218-220 on [K_init] vectors): one launch fewer per evaluation.
 *     out = log(sum_k exp(x_k - max_k x_k) + eps) + max_k x_k + add_const,   x_k = out_vec[k] + sum_f extra_f[k * stride_f]
 * written to *out, or through a result ring (ring_*, as alan_reduce_desc_t).  One chain (B = 1), fp32, 12 < K <= 32 (the
 * one-wave-per-product kernel); ALAN_ERR_UNSUPPORTED otherwise (nothing launched: pass fin = NULL and call alan_reduce). */
typedef struct {
    int32_t n_extra;                  /* 0 .. 3 further [K] factors */
    const void *extra[3];
    int64_t stride[3];                /* element strides (0 = a scalar broadcast) */
    double add_const;
    void *out;
    void *ring_slots, *ring_counter;
    int32_t ring_n;
} alan_chain_final_t;
int alan_chain_logmmexp_terms_final(const void *const *terms, const int64_t *strides, int32_t n_terms,
                                    const alan_chain_normal_t *normal, const alan_chain_final_t *fin, int32_t dtype,
                                    int64_t B, int64_t T, int64_t K, void *out_chain, void *out_vec, void *workspace,
                                    size_t workspace_bytes, void *stream);
size_t alan_chain_backward_batched_workspace_bytes(int64_t B, int64_t T, int64_t K, int32_t dtype);
int alan_chain_logmmexp_backward_batched(const void *ms, int32_t dtype, int64_t B, int64_t T, int64_t K,
                                         int64_t sB, int64_t sT, int64_t sRow, int64_t sCol,
                                         const void *tree, const void *out_vec, const void *grad_vec,
                                         const void *grad_chain, void *grad_ms,
                                         void *workspace, size_t workspace_bytes, void *stream);

/* Posterior sampling of a timeseries variable's K index at every timestep -- the role of sample_Ks_timeseries
 * (reduce_Ks.py:85-232, O(T^2) chain evaluations) as backward messages + forward sampling, two launches whatever T is.
 * ms is [C, T, K, K] fp32 with element strides (sC, sT, sRow, sCol): C chains of the [T, K_init, K] factor of
 * logpq.py:133.  K <= 128.
 *   alan_chain_messages   beta [C, T+1, K] contiguous:  beta[c,T,:] = 0,  beta[c,t,a] = LSE_b(ms[c,t,a,b] + beta[c,t+1,b])
 *   alan_chain_sample     out [N, B, T] int64:  k_t ~ softmax_b(ms[c,t,k_{t-1},b] + beta[c,t+1,b]) for t = 0..T-1, with
 *                         k_{-1} = init[n*iN + b*iB] (int64), chain c = n*cN + b*cB, one uniform in [0,1) per draw in
 *                         uniforms [N, B, T] fp32 (the caller's generator: draws are reproducible from its seed)
 *   alan_chain_filter     alpha [C, T, N, K] contiguous: the forward recursion from init[n] -- what the reference's
 *                         per-timestep draws are taken from (after mixing over n and normalising), for parity checks */
int alan_chain_messages(const void *ms, int64_t C, int64_t T, int64_t K, int64_t sC, int64_t sT, int64_t sRow,
                        int64_t sCol, void *beta, void *stream);
int alan_chain_sample(const void *ms, int64_t T, int64_t K, int64_t sC, int64_t sT, int64_t sRow, int64_t sCol,
                      const void *beta, const void *init, int64_t iN, int64_t iB, const void *uniforms, int64_t N,
                      int64_t B, int64_t cN, int64_t cB, void *out, void *stream);
int alan_chain_filter(const void *ms, int64_t C, int64_t T, int64_t K, int64_t sC, int64_t sT, int64_t sRow,
                      int64_t sCol, const void *init, int64_t N, void *alpha, void *stream);

/* One-shot sum of the ranks' partial log-marginals (the collective of a sharded Split: logpq.py:149-153 evaluated by
 * `world` ranks, one per GPU of a node, each over its own chunks of the plate).  Every rank writes its partial into a
 * slot of every peer's inbox over xGMI, raises a flag there, waits (for a bounded time) for the peers' flags in its own
 * inbox and adds the slots in rank order: ONE launch of `world` workgroups per exchange, the same bits on every rank,
 * in place of RCCL's all-reduce.  The exchange's running number lives on the device, so a launch captured into a HIP
 * graph advances it on every replay; every rank must issue the same sequence of exchanges.
 *   alan_exchange_create   allocates this rank's inbox on the current device (the ONE place this library owns device
 *                          memory: an inbox must be a whole allocation to be exported) and writes its HIP IPC handle
 *                          (ALAN_EXCHANGE_HANDLE_BYTES) to handle_out; capacity = the largest n, in fp32 elements.
 *                          Uncached device memory; the environment variable ALAN_EXCHANGE_ALLOC=plain asks for hipMalloc.
 *   alan_exchange_connect  handles = the world handles, rank-major (the caller carries them between the processes, e.g.
 *                          with torch.distributed.all_gather_object); opens the peers' inboxes.  Once.
 *   alan_exchange_sum      out[0..n) = sum over ranks q = 0..world-1, in that order, of rank q's src[0..n); fp32, device
 *                          pointers, enqueued on `stream`.  A peer that has not delivered within ALAN_EXCHANGE_SPIN_MS
 *                          (environment, default 2000) ends the wait: out is filled with NaN and the exchange is marked
 *                          failed (alan_exchange_status) -- no launch spins for ever.  A rank that has failed once
 *                          STAYS failed: every later exchange on it yields NaN as well (its peers may have completed
 *                          the failed exchange with a valid sum: the ranks have diverged, and the failed one says so
 *                          loudly) -- destroy and re-create the exchange on every rank behind a barrier.  Issue the
 *                          first exchange only after every rank has connected (a host barrier).
 *   alan_exchange_status   synchronous read-back: exchanges completed on this rank, and the number of the first one
 *                          that timed out (0 = none).
 *   alan_exchange_destroy  closes the peers' inboxes and frees this rank's (after the peers have stopped writing). */
#define ALAN_EXCHANGE_MAX_RANKS 8
#define ALAN_EXCHANGE_HANDLE_BYTES 64
int alan_exchange_create(int32_t world, int32_t rank, int32_t capacity, unsigned char *handle_out, void **exchange);
int alan_exchange_connect(void *exchange, const unsigned char *handles);
int alan_exchange_sum(void *exchange, const void *src, void *out, int64_t n, void *stream);
int alan_exchange_status(void *exchange, uint32_t *completed, uint32_t *failed_at);
int alan_exchange_destroy(void *exchange);

/* A recorded sequence of this library's launches, issued again in order by ONE call on the given stream: for a caller that
 * evaluates the same contraction again and again (Sample.elbo_nograd on a fixed sample -- logpq.py:68-155 once per call of
 * the reference's runner loop).  Measured against replaying the same launches as a captured HIP graph on MI355X: 23 us per
 * movielens K=30 evaluation instead of 27 (a graph launch leaves the GPU idle for several microseconds between two
 * replays; launches issued one by one do not).
 *   alan_calls_begin   from now until alan_calls_end, every entry point of this library called BY THIS THREAD plans its
 *                      call exactly as usual and keeps its kernel launches -- kernel, grid, a copy of every argument (host
 *                      memory of the library's own) -- in `calls` WITHOUT issuing them.  One list at a time per thread.
 *   alan_calls_end     ALAN_ERR_UNSUPPORTED if some launch could not be kept (a descriptor carrying timing events): the
 *                      list must not be replayed.
 *   alan_calls_count   kernel launches the list holds (what a caller compares with the node count of a captured graph
 *                      of the same evaluation, to know that the list IS the evaluation).
 *   alan_calls_replay  issues the kept launches in order and plans nothing.
 * Every device pointer of the recorded calls, workspaces included, must stay valid for as long as the list is replayed. */
int alan_calls_create(void **calls);
int alan_calls_begin(void *calls);
int alan_calls_end(void *calls);
int64_t alan_calls_count(void *calls);
int alan_calls_replay(void *calls, void *stream);
int alan_calls_destroy(void *calls);
/* Copies one generator slot {counter, seed} to another (see alan_noise_t). */
int alan_noise_handon(const void *from, void *to, void *stream);

/* INDEPENDENT evaluations overlapped.  Consecutive ELBO evaluations of the reference's loop (basic_runner.py:81-112;
 * logpq.py:68-155 per evaluation) do not depend on each other, while ONE evaluation is a chain of dependent launches most
 * of which occupy a fraction of the chip.  A pipeline holds n_lanes recorded copies of the evaluation (each alan_calls
 * list recorded over intermediates and outputs of its own) and issues evaluation i on lane i % n_lanes, every lane on a
 * stream the pipeline owns, so that one evaluation's small launches run beside another's plate step.  n_threads issuing
 * threads of the library's own (0: alan_pipeline_submit issues from the caller's thread) -- with the chip kept busy the
 * host's launch calls are the bound, and threads issuing to different streams do not serialise.
 *   alan_pipeline_submit  `count` more evaluations; returns at once (n_threads > 0).
 *   alan_pipeline_join    returns when every submitted evaluation has been ISSUED, having made `stream` wait for all of
 *                         them: work enqueued on `stream` afterwards sees every result.  Does not synchronise the device.
 *   alan_pipeline_fence   the lanes wait for everything enqueued on `stream` so far -- after the caller changed, on that
 *                         stream, memory the evaluations read (an optimiser step, new particles).
 * The one place beside alan_exchange_* where the library owns resources (streams, events, threads). */
#define ALAN_PIPELINE_MAX_LANES 8
int alan_pipeline_create(void *const *calls, int32_t n_lanes, int32_t n_threads, void **pipeline);
int alan_pipeline_submit(void *pipeline, int64_t count);
int alan_pipeline_join(void *pipeline, void *stream);
int alan_pipeline_fence(void *pipeline, void *stream);
int alan_pipeline_destroy(void *pipeline);

/* The optimiser step of the reference's training loop (basic_runner.py:108-110: opt.step() of a torch.optim.Adam) for up to
 * ALAN_ADAM_MAX_TENSORS fp32 parameter tensors in ONE launch, the step count on the device: a training iteration is then
 * library launches from the draws to the update and can be recorded and re-issued like an evaluation (alan_calls_*).
 * Arithmetic: Adam as torch.optim.Adam(capturable=True, fused=True) evaluates it (no weight decay, no amsgrad):
 *     m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g g;  p -= (lr / (1 - beta1^t)) m / (sqrt(v / (1 - beta2^t)) + eps)
 * with g -> -g when `maximize`.  All tensors contiguous fp32 of numel[i] elements.
 *   step    device float: the number of steps taken so far (0 before the first); this launch uses step + 1 and stores it
 *   ticket  device int32, ZERO before the first call and left zero by every launch (the last workgroup to finish advances
 *           `step`); one per optimiser, calls on it ordered by the stream */
#define ALAN_ADAM_MAX_TENSORS 24
typedef struct {
    int32_t n_tensors, maximize;
    void *param[ALAN_ADAM_MAX_TENSORS];
    const void *grad[ALAN_ADAM_MAX_TENSORS];
    void *exp_avg[ALAN_ADAM_MAX_TENSORS], *exp_avg_sq[ALAN_ADAM_MAX_TENSORS];
    int64_t numel[ALAN_ADAM_MAX_TENSORS];
    double lr, beta1, beta2, eps;
    void *step, *ticket;
} alan_adam_desc_t;
int alan_adam_step(const alan_adam_desc_t *desc, void *stream);

/* Library/ABI version and the gfx target it was built for (e.g. "gfx950"). */
int alan_abi_version(void);
const char *alan_build_target(void);

#ifdef __cplusplus
}
#endif
#endif /* ALAN_MI355_H */
