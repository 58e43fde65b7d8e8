// bpp_stats.inl -- episode statistics and the host hand-over of a step's finished bins, included by bpp_kernels.hip inside its
// anonymous namespace (round 6: moved out of that file unchanged): the fixed-order reductions of the per-bin accumulator rows
// (bpp_episode_stats, bpp_episode_acc_reduce; main.py:159-162), the ordered compaction of the finished bins' infos
// (bpp_gather_finished; bench/monitor.py:64-75) and the marker kernel of bpp_mark.  Host entry points: bpp_kernels.hip.

// Fixed-order reductions of the episode statistics (include/bpp_abi.h: BPP_REDUCE_LANES partial sums over strided
// bins, then a binary tree; ONE workgroup, so the order -- and with it every bit of the four float64 sums that
// multi-GPU jobs all-reduce -- is the same on every run and equals the oracle's).
__device__ __forceinline__ void reduce_tree_1024(double s0, double s1, double s2, double s3, double *acc) {
    static __shared__ double part[4][BPP_REDUCE_LANES];
    const int t = threadIdx.x;
    part[0][t] = s0;
    part[1][t] = s1;
    part[2][t] = s2;
    part[3][t] = s3;
    __syncthreads();
    for (int d = BPP_REDUCE_LANES / 2; d > 0; d >>= 1) {
        if (t < d) {
#pragma unroll
            for (int k = 0; k < 4; ++k) part[k][t] = part[k][t] + part[k][t + d];
        }
        __syncthreads();
    }
    if (t < 4) acc[t] = acc[t] + part[t][0];
}

// Stand-alone episode statistics (main.py:159-162) for callers that do not use bpp_batch.ep_acc.
__global__ __launch_bounds__(BPP_REDUCE_LANES) void stats_kernel(const uint8_t *done, const double *ep_ret, const double *ratio,
                                                                 const int32_t *ep_len, int E, double *acc) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int e = threadIdx.x; e < E; e += BPP_REDUCE_LANES)
        if (done[e]) {
            s0 = s0 + ep_ret[e];
            s1 = s1 + ratio[e];
            s2 = s2 + (double)ep_len[e];
            s3 = s3 + 1.0;
        }
    reduce_tree_1024(s0, s1, s2, s3, acc);
}

// The same from the per-bin accumulator rows bpp_step keeps (bpp_batch.ep_acc).
__global__ __launch_bounds__(BPP_REDUCE_LANES) void acc_reduce_kernel(double *ep_acc, int E, double *acc, int clear) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    // rows threadIdx.x, + 1024, + 2048, ... added in that order; eight rows are in flight at a time (the loads are
    // independent, only the additions are ordered)
    constexpr int U = 8;
    int e = threadIdx.x;
    for (; e + (U - 1) * BPP_REDUCE_LANES < E; e += U * BPP_REDUCE_LANES) {
        double v[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double *a = (const double *)__builtin_assume_aligned(ep_acc + 4 * (size_t)(e + u * BPP_REDUCE_LANES), 32);
            v[u][0] = a[0], v[u][1] = a[1], v[u][2] = a[2], v[u][3] = a[3];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s0 = s0 + v[u][0];
            s1 = s1 + v[u][1];
            s2 = s2 + v[u][2];
            s3 = s3 + v[u][3];
            if (clear) {
                double *a = (double *)__builtin_assume_aligned(ep_acc + 4 * (size_t)(e + u * BPP_REDUCE_LANES), 32);
                a[0] = 0.0, a[1] = 0.0, a[2] = 0.0, a[3] = 0.0;
            }
        }
    }
    for (; e < E; e += BPP_REDUCE_LANES) {
        double *a = (double *)__builtin_assume_aligned(ep_acc + 4 * (size_t)e, 32);
        const double v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3];
        s0 = s0 + v0;
        s1 = s1 + v1;
        s2 = s2 + v2;
        s3 = s3 + v3;
        if (clear) a[0] = 0.0, a[1] = 0.0, a[2] = 0.0, a[3] = 0.0;
    }
    reduce_tree_1024(s0, s1, s2, s3, acc);
}

// The same reduction spread over the chip (bpp_episode_acc_reduce with a scratch buffer).  The normative order has 1 024
// partial sums, each ONE sequential chain over its strided rows -- so 1 024 lanes is all the parallelism there is, and in
// one workgroup they share one CU's memory pipeline (2 MB at ~30 GB/s).  Here every workgroup owns kAccWideLanes of the
// partials (16 consecutive rows = one 512-byte line group per load instruction), fetches kAccWideRows rows of each at
// once with all its threads, publishes its partials to the caller's scratch buffer and takes a ticket; the last arriver runs the binary
// tree over all 1 024 partials.  Hand-off per MI355X_MICROARCH.md (inter-workgroup visibility): plain stores ->
// __syncthreads -> lane-0 agent-scope release -> s_waitcnt vmcnt(0) -> relaxed agent atomic; consumer: agent-scope
// acquire behind the ticket -> __syncthreads -> plain loads.
#ifndef BPP_DRAIN_VMEM   // (the host emulator of tests/emu defines it away: there is no vector memory queue to drain)
#define BPP_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")   // invisible to the compiler's waitcnt pass, which may drop its own
#endif
constexpr int kAccWideLanes = 16, kAccWideGroups = BPP_REDUCE_LANES / kAccWideLanes, kAccWideRows = 64;
__global__ __launch_bounds__(256) void acc_reduce_wide_kernel(double *ep_acc, int E, double *acc, int clear, double *scratch) {
    static __shared__ double part[4][BPP_REDUCE_LANES];
    static __shared__ int last;
    const int t = threadIdx.x;
    unsigned int *ticket = (unsigned int *)(scratch + 4 * BPP_REDUCE_LANES);
    // All 256 threads fetch: thread (j, l) = (t / 16, t % 16) brings rows j, j + 16, j + 32, j + 48 of a chunk of 64 rows of
    // partial l into LDS (`part` is free until the tree) -- one memory latency per chunk instead of one per kAccWideU rows
    // of a lane's own chain; then 64 threads, one per (partial, component), add the chunk's 64 values IN ROW ORDER: the same
    // additions in the same order as the one-workgroup kernel (a row beyond E is read as +0.0, which changes no sum that
    // started from +0.0).
    {
        double (*rows)[kAccWideLanes][4] = (double (*)[kAccWideLanes][4]) & part[0][0];     // [64][16][4] = 32 KB
        const int l = t % kAccWideLanes, j = t / kAccWideLanes;
        const int r = blockIdx.x * kAccWideLanes + l;
        const int al = t / 4 % kAccWideLanes, ak = t % 4;      // adder thread t < 64: partial al, component ak
        double sum = 0.0;
        for (int e0 = 0; e0 < E; e0 += kAccWideRows * BPP_REDUCE_LANES) {
            double v[kAccWideRows / 16][4];
#pragma unroll
            for (int m = 0; m < kAccWideRows / 16; ++m) {
                const int e = e0 + (j + 16 * m) * BPP_REDUCE_LANES + r;
                v[m][0] = v[m][1] = v[m][2] = v[m][3] = 0.0;
                if (e < E) {
                    double *a = (double *)__builtin_assume_aligned(ep_acc + 4 * (size_t)e, 32);
                    v[m][0] = a[0], v[m][1] = a[1], v[m][2] = a[2], v[m][3] = a[3];
                    if (clear) a[0] = 0.0, a[1] = 0.0, a[2] = 0.0, a[3] = 0.0;
                }
            }
#pragma unroll
            for (int m = 0; m < kAccWideRows / 16; ++m)
#pragma unroll
                for (int k = 0; k < 4; ++k) rows[j + 16 * m][l][k] = v[m][k];
            __syncthreads();
            if (t < 4 * kAccWideLanes) {
#pragma unroll 8
                for (int q = 0; q < kAccWideRows; ++q) sum = sum + rows[q][al][ak];
            }
            __syncthreads();
        }
        if (t < 4 * kAccWideLanes) scratch[ak * BPP_REDUCE_LANES + blockIdx.x * kAccWideLanes + al] = sum;
    }
    __syncthreads();
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        BPP_DRAIN_VMEM();
        const unsigned int n = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = n == (unsigned int)(kAccWideGroups - 1);
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!last) return;
    for (int i = t; i < 4 * BPP_REDUCE_LANES; i += 256) (&part[0][0])[i] = scratch[i];
    __syncthreads();
    for (int d = BPP_REDUCE_LANES / 2; d > 0; d >>= 1) {
        for (int r = t; r < d; r += 256) {
#pragma unroll
            for (int k = 0; k < 4; ++k) part[k][r] = part[k][r] + part[k][r + d];
        }
        __syncthreads();
    }
    if (t < 4) acc[t] = acc[t] + part[t][0];
    if (t == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next (stream-ordered) call
}

// bpp_gather_finished: ordered compaction of the finished bins' per-bin outputs.  Workgroup w owns the bins [w * chunk, (w + 1) *
// chunk) (chunk a multiple of 4 096; at most 64 workgroups) and needs no word from the others: it counts the finished bins in
// front of its chunk itself (the `done` bytes before it: at most 64 KB per 65 536 bins, 16 bytes per load, resident in L2), then
// compacts its own bins in rounds of 4 096 -- thread t of a round owns 16 consecutive bins; exclusive prefix of the per-thread
// counts by wave shuffles + one LDS pass over the 4 waves.  One launch, no scratch, no hand-off between workgroups, output in
// ascending bin order whatever the order the workgroups run in.  (Round 4 ran ONE workgroup of 1 024 threads over all bins: four
// serial rounds of gathers at 65 536 bins, ~60 us on the device.)  Output: header + five arrays of n entries (include/bpp_abi.h);
// entries beyond n (a caller whose count is wrong) are dropped, the header tells.
constexpr int kGatherThreads = 256, kGatherRound = kGatherThreads * 16, kGatherMaxGroups = 64;
__device__ __forceinline__ int nonzero_bytes(uint32_t v) {
    const uint32_t t = ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v;    // bit 7 of every byte that is not zero
    return __popc(t & 0x80808080u);
}
// bpp_mark where the runtime has no stream memory operation: one thread, one system-scope store
__global__ void mark_kernel(uint32_t *flag, uint32_t value) { __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

__global__ __launch_bounds__(kGatherThreads) void compact_finished_kernel(const uint8_t *done, const double *ep_ret, const double *ratio,
                                                                          const int32_t *ep_len, const int32_t *counter, int E,
                                                                          unsigned char *out, int n, int chunk) {
    constexpr int NW = kGatherThreads / 64;
    static __shared__ int wave_tot[NW];
    static __shared__ int wave_off[NW];
    static __shared__ int total;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double *o_ret = (double *)(out + 32), *o_ratio = o_ret + n;
    int32_t *o_len = (int32_t *)(o_ratio + n), *o_cnt = o_len + n, *o_bin = o_cnt + n;
    const int c_lo = (int)blockIdx.x * chunk, c_hi = min(E, c_lo + chunk);
    const bool aligned = (((uintptr_t)done) & 15u) == 0;
    // ---- finished bins in front of this workgroup's chunk (c_lo is a multiple of 4 096)
    int before = 0;
    if (aligned) {
        for (int i = t; i < c_lo / 16; i += kGatherThreads) {
            const uint4 v = ((const uint4 *)done)[i];
            before += nonzero_bytes(v.x) + nonzero_bytes(v.y) + nonzero_bytes(v.z) + nonzero_bytes(v.w);
        }
    } else {
        for (int i = t; i < c_lo; i += kGatherThreads) before += done[i] != 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d, 64);
    if (lane == 0) wave_tot[wave] = before;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) base += wave_tot[k];
    __syncthreads();
    // ---- this workgroup's own bins
    for (int c0 = c_lo; c0 < c_hi; c0 += kGatherRound) {
        const int e0 = c0 + t * 16;
        uint32_t m = 0;
        if (e0 + 16 <= c_hi && aligned) {
            const uint4 v = *(const uint4 *)(done + e0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) m |= ((w[q] >> (8 * k)) & 255u) ? 1u << (4 * q + k) : 0u;
        } else {
            for (int k = 0; k < 16; ++k)
                if (e0 + k < c_hi && done[e0 + k]) m |= 1u << k;
        }
        const int cnt = __popc(m);
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        if (t == 0) {
            int s = 0;
            for (int k = 0; k < NW; ++k) {
                wave_off[k] = s;
                s += wave_tot[k];
            }
            total = s;
        }
        __syncthreads();
        int pos = base + wave_off[wave] + incl - cnt;
        while (m) {
            const int k = __ffs((int)m) - 1;
            m &= m - 1;
            const int e = e0 + k;
            if (pos < n) {
                o_ret[pos] = ep_ret[e];
                o_ratio[pos] = ratio[e];
                o_len[pos] = ep_len[e];
                o_cnt[pos] = counter[e];
                o_bin[pos] = e;
            }
            ++pos;
        }
        base += total;
        __syncthreads();
    }
    if (blockIdx.x == gridDim.x - 1 && t < 8) ((int32_t *)out)[t] = t == 0 ? base : 0;   // the last chunk's running count is the total
}
