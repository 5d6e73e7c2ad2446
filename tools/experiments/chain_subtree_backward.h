// EXPERIMENT RECORD (round 4, VERDICT r3 item 5 "a workgroup per subtree") -- not part of the library.
// Three rounds of the timeseries tree's backward per workgroup, gradients handed down through LDS; parity-green
// (tests/test_gpu_chain_batched.py, test_gpu_reduce.py -k chain, test_timeseries.py: 86 passed) and SLOWER than a workgroup per
// node (csrc/chain.hip: chain_tree_backward_kernel): T = 1000, K = 30: 74.0 us (rounds per group 3), 85.0 (2), 119.2 (1)
// against 66.2.  rocprofv3 of the kernel alone: T = 4 (one group, two rounds) 22.9 us, T = 8 (three rounds) 31.9, T = 64
// (two levels of groups) 50.3, T = 512 68.3, T = 1000 71.1 -- a round INSIDE a group costs 5-9 us: its four dependent phases
// (row / column sums of G', the 15 dependent matrix instructions of a product, the epilogue's divisions into the children's
// LDS, two workgroup barriers over 14 waves) run on two waves with nothing to hide their latencies behind, on one CU per
// seven nodes; the hand-over through memory it saves is 3 us.  (Before an opaque copy of K kept the epilogue's sixteen 64-bit
// store addresses from being hoisted out of the rounds' loop the kernel spilled 288 bytes per thread: 106 us.)
// What follows is the kernel and its launch as they were in chain.hip (it needs that file's BwdLevel, BWD_UNSET, Num<>).

// K <= 32: THREE rounds of the tree per workgroup ("a workgroup per subtree").  A workgroup holds a node,
// its two children and its four grandchildren -- two waves each, one per product -- and hands the gradients down through
// LDS: of the nine hand-overs through memory of a T = 1000 tree three are left.  The groups' levels are laid out from the
// leaves up (the top group takes what is left: 1-3 rounds), a group's index is above its parent group's.  Every node of a
// group does its own half (stage, maxima, exponentials, ties, Pe @ Ce + eps) at once; then the rounds follow one another
// inside the workgroup, a parent's epilogue writing G / (Pe @ Ce + eps) straight into its children's LDS.
constexpr int ST_NODES = 7, ST_THREADS = 128 * ST_NODES, ST_MAX_GL = 12;
struct SubTree {
    uint32_t gfirst[ST_MAX_GL];       // first workgroup of group level g; 0xffffffff beyond the last
    int32_t glevel[ST_MAX_GL];        // the tree level (0 = the root's round) of a group's top node
    int32_t gdepth[ST_MAX_GL];        // rounds a group of this level covers (1 .. 3)
    BwdLevel lv[BWD_MAX_LEVELS];
    uint32_t timeout;
    int32_t K;
    const float *root;
    int64_t rB, rRow, rCol;
    const float *vec, *grad_vec, *grad_chain;
};

__global__ __launch_bounds__(ST_THREADS) void chain_subtree_backward_kernel(const SubTree a) {
    typedef float T;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int K = a.K, KS = K | 1, KK = K * K;
    const int per_node = 3 * K * KS + 6 * K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (scalar: the node, its level and every address that depends on them alone)
    const int p = wave >> 1, half = wave & 1, tl = (half << 6) | lane;        // local node, its wave, thread within the node
    constexpr int NTN = 128;
    const uint32_t vb = blockIdx.x;
    const int64_t b = blockIdx.y;
    int g = 0;
#pragma unroll
    for (int i = 1; i < ST_MAX_GL; ++i) g += vb >= a.gfirst[i] ? 1 : 0;
    typedef __attribute__((address_space(4))) const char *kernarg_ptr;
    const char *base = (const char *)((kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr());
    const uint32_t gfirst = *reinterpret_cast<const uint32_t *>(base + offsetof(SubTree, gfirst) + (size_t)g * 4);
    const int li0 = *reinterpret_cast<const int32_t *>(base + offsetof(SubTree, glevel) + (size_t)g * 4);
    const int depth = *reinterpret_cast<const int32_t *>(base + offsetof(SubTree, gdepth) + (size_t)g * 4);
    const int jtop = (int)(vb - gfirst);
    // this thread's node: depth d in the group, the o-th of that depth
    const int d = p >= 3 ? 2 : p >= 1 ? 1 : 0, o = p - ((1 << d) - 1);
    const int li = li0 + min(d, depth - 1);                                   // (clamped: a node beyond the group's depth is off)
    const BwdLevel &lv = *reinterpret_cast<const BwdLevel *>(base + offsetof(SubTree, lv) + (size_t)li * sizeof(BwdLevel));
    const int node = (jtop << d) + o;
    const bool on = d < depth && node < lv.n_nodes;
    const int n_src = lv.n_src, t0 = 2 * node, t1 = t0 + 1;
    const bool pair = on && t1 < n_src;
    const bool inner = d + 1 < depth;                                         // its children are nodes of this group
    T *mine = reinterpret_cast<T *>(smem_raw) + (size_t)p * per_node;
    T *Pe = mine, *Ce = Pe + K * KS, *Gp = Ce + K * KS, *pm = Gp + K * KS, *cm = pm + K, *pw = cm + K, *cw = pw + K, *pn = cw + K,
      *cn = pn + K;
    auto node_mem = [&](int q) { return reinterpret_cast<T *>(smem_raw) + (size_t)q * per_node; };
    const T NINF = Num<T>::ninf();
    const float invK = 1.f / (float)K;
    auto row_of = [&](int e) {
        int i = (int)((float)e * invK);
        i -= (i * K > e) ? 1 : 0;
        i += ((i + 1) * K <= e) ? 1 : 0;
        return i;
    };
    auto upstream_root = [&](int i, int j) -> T {
        T gr = a.grad_chain ? a.grad_chain[(b * K + i) * K + j] : T(0);
        if (a.grad_vec) {
            const T v = a.vec[b * K + i];
            if (v != NINF) gr += a.grad_vec[b * K + i] * Num<T>::exp_acc(a.root[b * a.rB + i * a.rRow + j * a.rCol] - v);
        }
        return gr;
    };
    const bool leaf_out = lv.leaf != 0;
    auto get = [&](const T *q) {
        T v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__float_as_uint(v) == BWD_UNSET) {
            const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
            do {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while (__float_as_uint(v) == BWD_UNSET && __builtin_amdgcn_s_memrealtime() - t_start <= a.timeout);
            if (__float_as_uint(v) == BWD_UNSET) v = __builtin_nanf("");
        }
        return v;
    };
    // ---- every node's own half
    if (pair) {
        const T *Pg = lv.src + b * lv.cB + (int64_t)t0 * lv.cT, *Cg = lv.src + b * lv.cB + (int64_t)t1 * lv.cT;
        const int64_t sRow = lv.cR, sCol = lv.cC;
        constexpr int CH = 4;
        for (int e0 = tl; e0 < KK; e0 += NTN * CH) {
            T pv[CH], cv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int e = min(e0 + NTN * u, KK - 1), i = row_of(e), j = e - i * K;
                pv[u] = Pg[i * sRow + j * sCol];
                cv[u] = Cg[i * sRow + j * sCol];
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int e = e0 + NTN * u;
                if (e < KK) {
                    const int i = row_of(e), j = e - i * K;
                    Pe[i * KS + j] = pv[u], Ce[i * KS + j] = cv[u];
                }
            }
        }
    }
    __syncthreads();
    if (pair) {
        for (int rc = tl; rc < 2 * K; rc += NTN) {
            const bool is_row = rc < K;
            const int idx = is_row ? rc : rc - K;
            const T *p0 = is_row ? Pe + idx * KS : Ce + idx;
            const int st = is_row ? 1 : KS;
            T m0 = NINF, m1 = NINF;
            int x = 0;
            for (; x + 2 <= K; x += 2) m0 = fmaxf(m0, p0[x * st]), m1 = fmaxf(m1, p0[(x + 1) * st]);
            for (; x < K; ++x) m0 = fmaxf(m0, p0[x * st]);
            (is_row ? pm : cm)[idx] = fmaxf(m0, m1);
        }
    }
    __syncthreads();
    if (pair) {
        for (int e = tl; e < KK; e += NTN) {
            const int i = row_of(e), j = e - i * K;
            Pe[i * KS + j] = Num<T>::exp_acc(Pe[i * KS + j] - pm[i]);
            Ce[i * KS + j] = Num<T>::exp_acc(Ce[i * KS + j] - cm[j]);
        }
    }
    __syncthreads();
    // rows of P / columns of C in four segments, a thread each (lanes 4 q .. 4 q + 3 of a wave: two shuffles add them up)
    auto walk4 = [&](auto f) {
        for (int q0 = 0; q0 < 2 * K; q0 += NTN / 4) {       // (uniform trip count)
            const int rc = q0 + (tl >> 2), seg = tl & 3;
            const bool in = rc < 2 * K;
            const bool is_row = rc < K;
            const int idx = in ? (is_row ? rc : rc - K) : 0;
            const int per = (K + 3) >> 2, x0 = seg * per, x1 = min(K, x0 + per);
            float v = f(is_row, idx, x0, x1);
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            if (in && seg == 0) f(is_row, idx, v);
        }
    };
    const int c = lane & 31, h = lane >> 5;
    // one 32 x 32 tile (K <= 32) on the matrix cores: lane (c, h) gives A[row c][k = 2 s + h] and B[k][column c]
    auto tile = [&](auto a_at, auto b_at) {
        chain_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int cc = min(c, K - 1);
        const bool okc = c < K;
        constexpr int SU = 4;
        for (int k0 = 0; k0 < K; k0 += 2 * SU) {
            float av[SU], bv[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int kk = k0 + 2 * u + h, k = min(kk, K - 1);
                const bool okk = kk < K;                  // (selects, not products with 0: 0 x inf would be NaN)
                av[u] = okc && okk ? a_at(cc, k) : 0.f;
                bv[u] = okc && okk ? b_at(k, cc) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        return acc;
    };
    struct Ties {
        const T *Pe, *Ce;
        T *pn, *cn;
        int KS;
        __device__ float operator()(bool is_row, int idx, int x0, int x1) const {
            const T *e0 = is_row ? Pe + idx * KS : Ce + idx;
            const int st = is_row ? 1 : KS;
            float n = 0.f;
            for (int x = x0; x < x1; ++x) n += e0[x * st] == T(1) ? 1.f : 0.f;
            return n;
        }
        __device__ void operator()(bool is_row, int idx, float v) const { (is_row ? pn : cn)[idx] = v; }
    };
    struct Sums {
        const T *Gp;
        T *pw, *cw;
        const T *pn, *cn;
        int KS;
        __device__ float operator()(bool is_row, int idx, int x0, int x1) const {
            const T *g0 = is_row ? Gp + idx * KS : Gp + idx;
            const int st = is_row ? 1 : KS;
            float s = 0.f;
            for (int x = x0; x < x1; ++x) s += g0[x * st];
            return s;
        }
        __device__ void operator()(bool is_row, int idx, float v) const {
            (is_row ? pw : cw)[idx] = Num<T>::eps * v / (is_row ? pn : cn)[idx];
        }
    };
    if (pair) {
        walk4(Ties{Pe, Ce, pn, cn, KS});
        if (half == 0) {
            const chain_f32x16 acc = tile([&](int i, int k) { return Pe[i * KS + k]; }, [&](int k, int j) { return Ce[k * KS + j]; });
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (i < K && c < K) Gp[i * KS + c] = acc[r] + Num<T>::eps;
            }
        }
    }
    // ---- the group's top node: the gradient from outside (the caller's at the root, the parent group's in memory)
    const T *Gn = lv.G ? lv.G + (b * lv.n_nodes + node) * (int64_t)KK : nullptr;
    if (tid == 0 && Gn) {                                  // (thread 0 is of the top node: Gn is its own)
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        while (__float_as_uint(__hip_atomic_load(Gn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == BWD_UNSET) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t_start > a.timeout) break;
        }
    }
    __syncthreads();
    if (p == 0 && on) {
        constexpr int CH = 4;                             // (a thread's loads requested four at a time before any is looked at)
        for (int e0 = tl; e0 < KK; e0 += NTN * CH) {
            T gv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u)
                gv[u] = Gn ? __hip_atomic_load(Gn + min(e0 + NTN * u, KK - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T(0);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int e = e0 + NTN * u;
                if (e < KK) {
                    const int i = row_of(e), j = e - i * K;
                    const T g1 = !Gn ? upstream_root(i, j) : __float_as_uint(gv[u]) == BWD_UNSET ? get(Gn + e) : gv[u];
                    Gp[i * KS + j] = pair ? g1 / Gp[i * KS + j] : g1;         // (a leftover node keeps the gradient as it is)
                }
            }
        }
    }
    __syncthreads();
    // ---- the rounds of the group, one after the other; a parent's epilogue feeds its children's Gp
    // where element (i, j) of the gradient of source matrix `which` (0: P, 1: C) of this thread's node goes
    T *child[2] = {node_mem(min(2 * p + 1, ST_NODES - 1)), node_mem(min(2 * p + 2, ST_NODES - 1))};       // (beyond the group: never used)
    // (is that child a pair?  its sources are the level below's matrices 2 t, 2 t + 1 of n_src_child)
    const BwdLevel &lvc = *reinterpret_cast<const BwdLevel *>(base + offsetof(SubTree, lv) + (size_t)(inner ? li + 1 : li) * sizeof(BwdLevel));
    const bool child_pair[2] = {inner && 2 * t0 + 1 < lvc.n_src, inner && 2 * t1 + 1 < lvc.n_src};
    T *dP = lv.dsrc + (b * n_src + t0) * (int64_t)KK, *dC = dP + KK;
    // where the gradient of source matrix 0 (P) / 1 (C) of this thread's node goes -- decided once, outside the element loops:
    // 0 a child's LDS, divided by its Pe @ Ce + eps; 1 a child's LDS as it is (a leftover child); 2 memory, for another
    // group; 3 memory, the caller's grad_ms
    const int mode_of[2] = {inner ? (child_pair[0] ? 0 : 1) : (leaf_out ? 3 : 2), inner ? (child_pair[1] ? 0 : 1) : (leaf_out ? 3 : 2)};
    for (int dd = 0; dd < depth; ++dd) {
        const bool act = on && d == dd;
        // (the sixteen store addresses of the epilogue are loop-invariant: hoisted out of this loop as 64-bit values they spill
        // the kernel's registers; an opaque copy of K keeps them where they are used)
        int Kq = K;
        asm volatile("" : "+s"(Kq));
        if (act && pair) walk4(Sums{Gp, pw, cw, pn, cn, KS});
        __syncthreads();
        if (act && pair) {
            chain_f32x16 acc;
            if (half == 0)      // dP[i, k] : A[i][j] = G'[i][j], B[j][k] = Ce[k][j]
                acc = tile([&](int i, int j) { return Gp[i * KS + j]; }, [&](int j, int k) { return Ce[k * KS + j]; });
            else                // dC[k, j] : A[k][r] = Pe[r][k], B[r][j] = G'[r][j]
                acc = tile([&](int k, int r) { return Pe[r * KS + k]; }, [&](int r, int j) { return Gp[r * KS + j]; });
            // lane (c, h): register r is element (row (r & 3) + 8 (r >> 2) + 4 h, column c)
            float val[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = min((r & 3) + 8 * (r >> 2) + 4 * h, K - 1), cc = min(c, K - 1);
                const T e = (half == 0 ? Pe : Ce)[i * KS + cc];
                val[r] = e * acc[r] + (e == T(1) ? (half == 0 ? pw[i] : cw[cc]) : T(0));
            }
            auto each = [&](auto sink) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (i < K && c < K) sink(i, c, r);
                }
            };
            T *gp = child[half] + 2 * K * KS, *gm = half ? dC : dP;
            switch (mode_of[half]) {                      // (wave-uniform)
                case 0: each([&](int i, int j, int r) { gp[i * KS + j] = val[r] / gp[i * KS + j]; }); break;
                case 1: each([&](int i, int j, int r) { gp[i * KS + j] = val[r]; }); break;
                case 2: each([&](int i, int j, int r) { __hip_atomic_store(gm + i * Kq + j, val[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }); break;
                default: each([&](int i, int j, int r) { gm[i * Kq + j] = val[r]; }); break;
            }
        } else if (act) {       // leftover of its round (utils.py:488-495): the gradient passes through to its one source
            T *gp = child[0] + 2 * K * KS;
            for (int e = tl; e < KK; e += NTN) {
                const int i = row_of(e), j = e - i * K;
                const T v = Gp[i * KS + j];
                switch (mode_of[0]) {
                    case 0: gp[i * KS + j] = v / gp[i * KS + j]; break;
                    case 1: gp[i * KS + j] = v; break;
                    case 2: __hip_atomic_store(dP + e, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
                    default: dP[e] = v; break;
                }
            }
        }
        __syncthreads();
    }
}


// ---- in chain_backward_run, behind the launch of chain_unset_kernel:
/*
            static const int sub_knob = env_knob("ALAN_CHAIN_BWD_SUBTREE");             // ablation knob: 0 = a workgroup per node
            const size_t smem_s = (size_t)ST_NODES * (3 * (size_t)K * (size_t)(K | 1) + 6 * (size_t)K) * sizeof(float);
            if (K <= 32 && sub_knob != 0 && smem_s <= 160 * 1024) {
                // three rounds per workgroup, the groups' levels laid out from the leaves up (chain_subtree_backward_kernel)
                SubTree st;
                std::memset(&st, 0, sizeof(st));
                std::memcpy(st.lv, a.lv, sizeof(st.lv));
                const int gd = sub_knob == 1 || sub_knob == 2 ? sub_knob : 3;            // (rounds per group: 3; the knob: 1 or 2)
                const int top_depth = tl.L % gd == 0 ? gd : tl.L % gd, ngl = 1 + (tl.L - top_depth) / gd;
                if (ngl <= ST_MAX_GL) {
                    uint32_t groups = 0;
                    for (int gI = 0; gI < ST_MAX_GL; ++gI) st.gfirst[gI] = 0xffffffffu;
                    for (int gI = 0, lev = 0; gI < ngl; ++gI) {
                        st.gfirst[gI] = groups, st.glevel[gI] = lev, st.gdepth[gI] = gI == 0 ? top_depth : gd;
                        groups += (uint32_t)a.lv[lev].n_nodes;
                        lev += st.gdepth[gI];
                    }
                    st.timeout = a.timeout, st.K = a.K, st.root = a.root, st.rB = a.rB, st.rRow = a.rRow, st.rCol = a.rCol;
                    st.vec = a.vec, st.grad_vec = a.grad_vec, st.grad_chain = a.grad_chain;
                    if (smem_s > 64 * 1024 &&
                        hipFuncSetAttribute((const void *)chain_subtree_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)smem_s) != hipSuccess)
                        return ALAN_ERR_LAUNCH;
                    ALAN_LAUNCH(chain_subtree_backward_kernel, dim3(groups, (uint32_t)B), dim3(ST_THREADS), smem_s, stream, st);
                    return hipGetLastError() == hipSuccess ? ALAN_OK : ALAN_ERR_LAUNCH;
                }
            }
*/
