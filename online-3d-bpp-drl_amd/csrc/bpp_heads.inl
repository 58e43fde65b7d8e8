// bpp_heads.inl -- the action-selection kernels, included by bpp_kernels.hip inside its anonymous namespace (round 6: moved out
// of that file unchanged): the uniform-feasible samplers of the benchmark policy (bpp_sample_feasible), the masked categorical
// head of the policy (bpp_masked_act / bpp_masked_act_counter; acktr/distributions.py:71-84) and its training half
// (bpp_masked_evaluate / _backward; acktr/model.py:90-96).  Host entry points: bpp_kernels.hip.

// Sub-groups of 16 lanes per bin (4 bins per wave): each lane owns `per` consecutive float4 quads of
// the bin's mask row (16-byte loads), an inclusive scan inside the 16-lane row locates the pick-th set
// entry in index order.  pick = (hash >> 32) * count >> 32.
template <int PER>
__global__ __launch_bounds__(256) void sample_kernel(const float *mask, int64_t *actions, int E, int M,
                                                     int64_t env_id_base, uint64_t seed, uint64_t step) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid >> 4, sl = threadIdx.x & 15;
    const bool active = e < E;
    const float4 *m = (const float4 *)(mask + (size_t)(active ? e : 0) * M);
    const int nq = M >> 2;
    float4 q[PER];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int qi = sl * PER + k;
        q[k] = (active && qi < nq) ? m[qi] : make_float4(0.f, 0.f, 0.f, 0.f);
        cnt += (q[k].x != 0.f) + (q[k].y != 0.f) + (q[k].z != 0.f) + (q[k].w != 0.f);
    }
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        const int o = __shfl_up(incl, d, 16);
        if (sl >= d) incl += o;
    }
    const int total = __shfl(incl, 15, 16);
    if (!active) return;
    if (total == 0) {
        if (sl == 0) actions[e] = 0;
        return;
    }
    int pick = (int)__umulhi(mix32(mix32_base(seed, step), (uint32_t)(env_id_base + e)), (uint32_t)total);
    const int excl = incl - cnt;
    if (pick >= excl && pick < incl) {
        pick -= excl;
        int found = 0, c = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float v[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (v[t] != 0.f) {
                    if (c == pick) found = (sl * PER + k) * 4 + t;
                    ++c;
                }
        }
        actions[e] = found;
    }
}

// Masked categorical action selection (include/bpp_abi.h: bpp_masked_act; acktr/distributions.py:71-84,
// acktr/model.py:56-68).  16 lanes per bin = one DPP row, PER float4 quads of logits and mask per lane (lane sl owns quads
// sl, sl + 16, ...: every load instruction of a row is one contiguous 256-byte segment).  Round 6: the kernel is VALU-issue
// bound, not memory bound -- 65 536 rows of M = 100 are 16 waves per SIMD, and round 1's 680 instructions per wave
// (38 ds_bpermute shuffles with their address arithmetic, two IEEE divisions, logf, per-element range predicates) were 16.2 us
// for 53 MB.  Now: row maximum, softmax denominator, probability total, the inclusive scan of the CDF and the index
// reductions run on the DPP data path (row_ror / row_shr: one VALU instruction each, no LDS), the reciprocals and the
// logarithm are the hardware's (v_rcp_f32 / v_log_f32, 1 ulp: far inside the 5e-6 log-probability tolerance the torch
// reference is held to), a quad past the end of the row is a -inf logit instead of a predicate per element, the sampled
// entry is found by COUNTING the cumulative sums below the target, and the lane that owns the chosen entry writes the outputs
// (no broadcast of its probability): ~340 instructions per wave.
// row_ror:n rotates within every 16-lane row; row_shr:n shifts, lanes without a source keep `old`
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v, float old) {
    (void)old;
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v, int old) {
    (void)old;
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<0x128>(v, v)); v = fmaxf(v, dpp_f<0x124>(v, v)); v = fmaxf(v, dpp_f<0x122>(v, v)); v = fmaxf(v, dpp_f<0x121>(v, v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0x128>(v, v); v += dpp_f<0x124>(v, v); v += dpp_f<0x122>(v, v); v += dpp_f<0x121>(v, v);
    return v;
}
__device__ __forceinline__ int row16_min(int v) {
    v = min(v, dpp_i<0x128>(v, v)); v = min(v, dpp_i<0x124>(v, v)); v = min(v, dpp_i<0x122>(v, v)); v = min(v, dpp_i<0x121>(v, v));
    return v;
}
__device__ __forceinline__ int row16_isum(int v) {
    v += dpp_i<0x128>(v, v); v += dpp_i<0x124>(v, v); v += dpp_i<0x122>(v, v); v += dpp_i<0x121>(v, v);
    return v;
}
__device__ __forceinline__ float row16_scan(float v) {   // inclusive prefix sum along the row
    v += dpp_f<0x111>(v, 0.0f); v += dpp_f<0x112>(v, 0.0f); v += dpp_f<0x114>(v, 0.0f); v += dpp_f<0x118>(v, 0.0f);
    return v;
}

template <int PER, bool DET>
__global__ __launch_bounds__(256) void masked_act_kernel(const float *logits, const float *mask, int64_t *action,
                                                         float *log_prob, int E, int M, int64_t env_id_base,
                                                         uint64_t seed, uint64_t step, const uint64_t *seed_step) {
    if (seed_step != nullptr) seed = seed_step[0], step = seed_step[1];   // bpp_masked_act_counter: (seed, step) live in device memory
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int e = tid >> 4, sl = threadIdx.x & 15;
    const bool active = e < E;
    const size_t row = (size_t)(active ? e : 0) * M;
    const float4 *xq = (const float4 *)(logits + row), *mq = (const float4 *)(mask + row);
    const int nq = M >> 2;
    float4 xv[PER], mv[PER];
    bool in[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {      // every load of both operands is issued before any arithmetic
        const int qi = sl + 16 * k;
        in[k] = qi < nq;
        // a quad past the end of the row behaves like four entries that can never be chosen: logit -inf (probability 0 before
        // the floor), and the 1e-5 floor itself is switched off for it below
        xv[k] = in[k] ? xq[qi] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        mv[k] = in[k] ? mq[qi] : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    float z[PER][4];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        z[k][0] = xv[k].x - (1.0f - mv[k].x) * 14.0f;  // distributions.py:76-79
        z[k][1] = xv[k].y - (1.0f - mv[k].y) * 14.0f;
        z[k][2] = xv[k].z - (1.0f - mv[k].z) * 14.0f;
        z[k][3] = xv[k].w - (1.0f - mv[k].w) * 14.0f;
        mx = fmaxf(fmaxf(mx, fmaxf(z[k][0], z[k][1])), fmaxf(z[k][2], z[k][3]));
    }
    mx = row16_max(mx);
    const float mxl = mx * 1.44269504088896340736f;
    float part = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            z[k][t] = __builtin_amdgcn_exp2f(z[k][t] * 1.44269504088896340736f - mxl);   // exp(z - mx); exp2(-inf) = 0
            part += z[k][t];
        }
    const float inv_sum = __builtin_amdgcn_rcpf(row16_sum(part));
    float qtot[PER];
    float lane_tot = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const float floor_k = in[k] ? 1e-5f : 0.0f;       // distributions.py:79-80
        qtot[k] = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            z[k][t] = z[k][t] * inv_sum + floor_k;
            qtot[k] += z[k][t];
        }
        lane_tot += qtot[k];
    }
    const float tot = row16_sum(lane_tot);
    int a;
    if constexpr (DET) {     // dist.mode(): first index of the maximum
        float best = -1.0f;
        int best_i = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k)
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (z[k][t] > best) {
                    best = z[k][t];
                    best_i = (sl + 16 * k) * 4 + t;
                }
        const float rb = row16_max(best);
        a = row16_min(best == rb ? best_i : 0x7fffffff);
    } else {
        // inverse CDF at u * total, entries in index order (quad-row k, then lane): the chosen entry is the first one whose
        // inclusive cumulative sum exceeds the target = the NUMBER of entries whose cumulative sum does not (the sums of a
        // lane grow with the index; the handful of cases where float32 rounding makes a lane's start fall an ulp below its
        // predecessor's end move a draw by one entry whose cumulative sum equals the target to ~1e-7 -- inside the CDF
        // tolerance the kernel is held to)
        const float u = (float)(mix32(mix32_base(seed, step), (uint32_t)(env_id_base + (active ? e : 0))) >> 8) * (1.0f / 16777216.0f);
        const float target = u * tot;
        float base = 0.0f;
        int below = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const float incl = row16_scan(qtot[k]);
            float c = base + incl - qtot[k];
            int bk = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                c += z[k][t];
                bk += c <= target ? 1 : 0;
            }
            below += in[k] ? bk : 0;
            if (k + 1 < PER) base += row16_sum(qtot[k]);
        }
        a = min(row16_isum(below), M - 1);          // (every sum <= target: rounding at u ~ 1 -> last entry)
    }
    // the lane that owns entry `a` holds its probability: it writes both outputs
    const int aq = a >> 2;
    if (active && (aq & 15) == sl) {
        float pa = 0.0f;
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if ((aq >> 4) == k) {
                const int t = a & 3;
                pa = t == 0 ? z[k][0] : (t == 1 ? z[k][1] : (t == 2 ? z[k][2] : z[k][3]));
            }
        action[e] = a;
        if (log_prob) {
            const float eps = 1.1920928955078125e-7f;  // torch clamp_probs: finfo(float32).eps
            log_prob[e] = __logf(fminf(fmaxf(pa * __builtin_amdgcn_rcpf(tot), eps), 1.0f - eps));
        }
    }
}

// Same selection for rows the 16-lane kernel cannot take (M not a multiple of 4, or M > 512 such as the
// 20x20 bin with rotation, M = 800): one wave per bin, entry k lives in lane k % 64, chunk k / 64; the CDF
// walks the chunks in order with an inclusive wave scan per chunk.
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}
__global__ __launch_bounds__(256) void masked_act_kernel_generic(const float *logits, const float *mask, int64_t *action,
                                                                 float *log_prob, int E, int M, int64_t env_id_base,
                                                                 uint64_t seed, uint64_t step, int deterministic, const uint64_t *seed_step) {
    if (seed_step != nullptr) seed = seed_step[0], step = seed_step[1];
    const int lane = threadIdx.x & (kWave - 1);
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;  // whole waves leave; no block-level synchronisation below
    const float *x = logits + (size_t)e * M, *m = mask + (size_t)e * M;
    const int nchunk = (M + kWave - 1) / kWave;
    float mx = -INFINITY;
    for (int k = lane; k < M; k += kWave) mx = fmaxf(mx, x[k] - (1.0f - m[k]) * 14.0f);  // distributions.py:76-79
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, kWave));
    float part = 0.0f;
    for (int k = lane; k < M; k += kWave) part += expf(x[k] - (1.0f - m[k]) * 14.0f - mx);
    const float sum = wave_sum_f(part);
    float lane_tot = 0.0f, best = -1.0f;
    int best_i = 0;
    for (int k = lane; k < M; k += kWave) {
        const float pk = expf(x[k] - (1.0f - m[k]) * 14.0f - mx) / sum + 1e-5f;  // distributions.py:79-80
        lane_tot += pk;
        if (pk > best) {
            best = pk;
            best_i = k;
        }
    }
    const float tot = wave_sum_f(lane_tot);
    int a;
    float pa;
    if (deterministic) {  // dist.mode(): first index of the maximum
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const float ob = __shfl_xor(best, d, kWave);
            const int oi = __shfl_xor(best_i, d, kWave);
            if (ob > best || (ob == best && oi < best_i)) {
                best = ob;
                best_i = oi;
            }
        }
        a = best_i;
        pa = best;
    } else {
        const float u = (float)(mix32(mix32_base(seed, step), (uint32_t)(env_id_base + e)) >> 8) * (1.0f / 16777216.0f);
        const float target = u * tot;
        float base = 0.0f, pm = 0.0f;
        int cand = 0x7fffffff;
        for (int c = 0; c < nchunk; ++c) {  // wave-uniform trip count
            const int k = c * kWave + lane;
            const float pk = k < M ? expf(x[k] - (1.0f - m[k]) * 14.0f - mx) / sum + 1e-5f : 0.0f;
            float incl = pk;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const float o = __shfl_up(incl, d, kWave);
                if (lane >= d) incl += o;
            }
            if (cand == 0x7fffffff && k < M && base + incl > target) {
                cand = k;
                pm = pk;
            }
            base += __shfl(incl, kWave - 1, kWave);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const int oc = __shfl_xor(cand, d, kWave);
            const float op = __shfl_xor(pm, d, kWave);
            if (oc < cand) {
                cand = oc;
                pm = op;
            }
        }
        if (cand == 0x7fffffff) {  // rounding at u ~ 1: the last entry
            cand = M - 1;
            pm = expf(x[M - 1] - (1.0f - m[M - 1]) * 14.0f - mx) / sum + 1e-5f;
        }
        a = cand;
        pa = pm;
    }
    if (lane == 0) {
        action[e] = a;
        if (log_prob) {
            const float eps = 1.1920928955078125e-7f;
            log_prob[e] = logf(fminf(fmaxf(pa / tot, eps), 1.0f - eps));
        }
    }
}

// Training half of the masked policy head (acktr/distributions.py:71-101 as used by Policy.evaluate_actions,
// acktr/model.py:90-96): for the actions taken, one wave per bin computes
//   logp  = log(clamp(p[a]))            p = lx / sum(lx), lx = softmax(x - 14 (1 - mask)) + 1e-5   (dist.log_probs)
//   ent   = -sum_k p_k log(clamp(p_k))                                                           (dist.entropy())
//   bad   = sum_k softmax(x)_k (1 - mask_k)                                                      (row sum of `bx`)
// and the backward kernel the gradient of  g_logp * logp + g_ent * ent + g_bad * bad  with respect to the logits
// (clamp = torch's probs_to_logits clamp to [eps, 1 - eps], derivative 0 outside).
struct RowStats {
    float mq, ma, sq, sa, tot;   // maxima and denominators of the masked / plain softmax, sum of lx
};
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, kWave));
    return v;
}
__device__ __forceinline__ RowStats masked_row_stats(const float *x, const float *m, int M, int lane) {
    RowStats r;
    float mq = -INFINITY, ma = -INFINITY;
    for (int k = lane; k < M; k += kWave) {
        mq = fmaxf(mq, x[k] - (1.0f - m[k]) * 14.0f);
        ma = fmaxf(ma, x[k]);
    }
    r.mq = wave_max_f(mq);
    r.ma = wave_max_f(ma);
    float sq = 0.0f, sa = 0.0f;
    for (int k = lane; k < M; k += kWave) {
        sq += expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq);
        sa += expf(x[k] - r.ma);
    }
    r.sq = wave_sum_f(sq);
    r.sa = wave_sum_f(sa);
    float tot = 0.0f;
    for (int k = lane; k < M; k += kWave) tot += expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq) / r.sq + 1e-5f;
    r.tot = wave_sum_f(tot);
    return r;
}
constexpr float kProbEps = 1.1920928955078125e-7f;   // torch.finfo(float32).eps, probs_to_logits clamp

__global__ __launch_bounds__(256) void masked_eval_fwd_kernel(const float *logits, const float *mask, const int64_t *action,
                                                              float *logp, float *entropy, float *bad, int E, int M) {
    const int lane = threadIdx.x & (kWave - 1);
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float *x = logits + (size_t)e * M, *m = mask + (size_t)e * M;
    const RowStats r = masked_row_stats(x, m, M, lane);
    float h = 0.0f, b = 0.0f;
    for (int k = lane; k < M; k += kWave) {
        const float p = (expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq) / r.sq + 1e-5f) / r.tot;
        h -= p * logf(fminf(fmaxf(p, kProbEps), 1.0f - kProbEps));
        b += expf(x[k] - r.ma) / r.sa * (1.0f - m[k]);
    }
    h = wave_sum_f(h);
    b = wave_sum_f(b);
    if (lane == 0) {
        const int64_t a = action[e];
        const float pa = (a >= 0 && a < M) ? (expf(x[a] - (1.0f - m[a]) * 14.0f - r.mq) / r.sq + 1e-5f) / r.tot : kProbEps;
        logp[e] = logf(fminf(fmaxf(pa, kProbEps), 1.0f - kProbEps));
        entropy[e] = h;
        bad[e] = b;
    }
}

__global__ __launch_bounds__(256) void masked_eval_bwd_kernel(const float *logits, const float *mask, const int64_t *action,
                                                              const float *g_logp, const float *g_ent, const float *g_bad,
                                                              float *grad, int E, int M) {
    const int lane = threadIdx.x & (kWave - 1);
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float *x = logits + (size_t)e * M, *m = mask + (size_t)e * M;
    float *g = grad + (size_t)e * M;
    const RowStats r = masked_row_stats(x, m, M, lane);
    const int64_t a = action[e];
    const float gl = g_logp[e], ge = g_ent[e], gb = g_bad[e];
    // h_k = dLoss/dp_k
    auto hk = [&](int k, float p) {
        const bool inside = p > kProbEps && p < 1.0f - kProbEps;
        const float pc = fminf(fmaxf(p, kProbEps), 1.0f - kProbEps);
        float h = -ge * (logf(pc) + (inside ? p / pc : 0.0f));
        if (k == a) h += inside ? gl / pc : 0.0f;
        return h;
    };
    float c = 0.0f, b = 0.0f;
    for (int k = lane; k < M; k += kWave) {
        const float p = (expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq) / r.sq + 1e-5f) / r.tot;
        c += p * hk(k, p);
        b += expf(x[k] - r.ma) / r.sa * (1.0f - m[k]);
    }
    c = wave_sum_f(c);   // sum_j p_j h_j
    b = wave_sum_f(b);   // bad
    float v = 0.0f;      // sum_j q_j u_j,  u_j = (h_j - c) / tot
    for (int k = lane; k < M; k += kWave) {
        const float q = expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq) / r.sq;
        v += q * (hk(k, (q + 1e-5f) / r.tot) - c) / r.tot;
    }
    v = wave_sum_f(v);
    for (int k = lane; k < M; k += kWave) {
        const float q = expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq) / r.sq;
        const float u = (hk(k, (q + 1e-5f) / r.tot) - c) / r.tot;
        const float av = expf(x[k] - r.ma) / r.sa;
        g[k] = q * (u - v) + gb * av * ((1.0f - m[k]) - b);
    }
}

// Fallback for rows that are not a multiple of 4 floats or longer than 16 * 8 quads: one wave per bin.
__global__ __launch_bounds__(256) void sample_kernel_generic(const float *mask, int64_t *actions, int E, int M,
                                                             int64_t env_id_base, uint64_t seed, uint64_t step) {
    const int lane = threadIdx.x & (kWave - 1);
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= E) return;
    const float *m = mask + (size_t)e * M;
    const int per = (M + kWave - 1) / kWave;
    const int b = min(lane * per, M), en = min(b + per, M);
    int cnt = 0;
    for (int k = b; k < en; ++k) cnt += (m[k] != 0.0f);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int o = __shfl_up(incl, d, kWave);
        if (lane >= d) incl += o;
    }
    const int total = __shfl(incl, kWave - 1, kWave);
    if (total == 0) {
        if (lane == 0) actions[e] = 0;
        return;
    }
    int pick = (int)__umulhi(mix32(mix32_base(seed, step), (uint32_t)(env_id_base + e)), (uint32_t)total);
    const int excl = incl - cnt;
    if (pick >= excl && pick < incl) {
        pick -= excl;
        for (int k = b; k < en; ++k)
            if (m[k] != 0.0f) {
                if (pick == 0) {
                    actions[e] = k;
                    break;
                }
                --pick;
            }
    }
}

