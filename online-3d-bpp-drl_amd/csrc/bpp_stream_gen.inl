// bpp_stream_gen.inl -- endless CUT-2 item supply generated ON THE DEVICE, included by
// bpp_kernels.hip inside its anonymous namespace.
//
// The reference draws every episode's sequence from the worker's `random` stream, one after the other
// (envs/bpp0/mdCreator.py:147-166: MDlayerBoxCreator.reset() -> bin.reset() -> gen_benchmark()).  Here every bin
// owns an exact MT19937 (CPython's random.Random(seed0 + global bin id): init_by_array seeding, getrandbits-based
// randbelow) kept in device memory, and a refill kernel -- one lane per bin -- cuts as many new sequences as the
// bin has consumed since the last refill into the bin's ring of D pool rows.  The step kernels are unchanged: they
// index the ring like any pool (row = (episode mod D) * E + bin; bpp_batch.pool_mode = BPP_POOL_RING).
//
// The cutting algorithm below is a second, independent statement of mdCreator.py:59-138 (the oracle library keeps
// the plain-C one of include/bpp_gen.inl): same draws in the same order, so sequence k of a bin equals the k-th
// sequence `random.Random(seed0 + id)` yields through the reference creator.  It is written once for host and
// device: `Rng` supplies u32(), `Work` the pending-box list.

// pending box: a = x | y << 8 | z << 16, b = low | high << 8 | alive << 31   (sides and heights <= 255)
struct CutBox {
    uint32_t a, b;
};
constexpr uint32_t kAlive = 0x80000000u;

template <class Rng>
__host__ __device__ inline uint32_t getrandbits_below(Rng &rng, uint32_t n) {   // Random._randbelow_with_getrandbits(n), 0 < n < 2^32
    int k = 0;
    for (uint32_t v = n; v; v >>= 1) ++k;
    uint32_t x = rng.u32() >> (32 - k);
    while (x >= n) x = rng.u32() >> (32 - k);
    return x;
}

// ---- the counter-based generator of BPP_STREAM_RNG_COUNTER (normative definition: include/bpp_abi.h) -----------------
__host__ __device__ inline uint32_t ctr_fmix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x85ebca6bu;
    x ^= x >> 13;
    x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
struct CtrRng {
    uint32_t klo, khi, n;
    __host__ __device__ static CtrRng key(uint64_t seed0, uint64_t sid, uint32_t k) {
        uint32_t h = ctr_fmix32((uint32_t)seed0 + 0x9E3779B9u);
        h = ctr_fmix32(h ^ (uint32_t)(seed0 >> 32));
        h = ctr_fmix32(h ^ (uint32_t)sid);
        h = ctr_fmix32(h ^ (uint32_t)(sid >> 32));
        CtrRng c;
        c.klo = ctr_fmix32(h ^ k);
        c.khi = ctr_fmix32(c.klo + 0x7F4A7C15u + k);
        c.n = 0;
        return c;
    }
    __host__ __device__ uint32_t word(uint32_t a) const { return ctr_fmix32((klo + n * 0x9E3779B9u) ^ (khi + a * 0x85EBCA77u)); }
    __host__ __device__ uint32_t below(uint32_t lim) {      // Lemire's unbiased multiply-shift; one draw index per call
        uint32_t a = 0;
        uint64_t m = (uint64_t)word(a) * lim;
        if ((uint32_t)m < lim) {                            // probability lim / 2^32
            const uint32_t t = (0u - lim) % lim;
            while ((uint32_t)m < t) m = (uint64_t)word(++a) * lim;
        }
        n += 1;
        return (uint32_t)(m >> 32);
    }
};

// One CUT-2 sequence, every box side in [lo, hi].  `work` is the pending list (get / set by index, capacity >=
// W*L*H / lo^3 + 8), `vals` collects the cut boxes as x | y<<8 | z<<16 | base height<<24 and sorts them.  Returns the
// number of items.
//
// The reference walks `invalid_box` with a `for` loop while removing the box just split and appending its oversized
// parts (mdCreator.py:117-135): a removal slides the rest of the list one place to the left under the iterator, which
// therefore SKIPS the element that followed the removed one; appended parts are visited in the same pass; the outer
// `while True` starts over until the list is empty.  Here the list is never shifted: a split box is marked dead, the
// element after it (dead ones do not count) is skipped once, and the survivors are compacted between passes -- the
// same visiting order, hence the same draws, without the O(n) shift per split.
template <class Rng, class Work, class Vals>
__host__ __device__ inline int cut2_generate(Rng &rng, Work &work, Vals &vals, int W, int L, int H, int lo, int hi) {
    int nv = 0, tail = 0;
    work.set(tail++, CutBox{(uint32_t)W | ((uint32_t)L << 8) | ((uint32_t)H << 16), 0u | ((uint32_t)H << 8) | kAlive});
    for (;;) {
        bool skip = false;
        for (int i = 0; i < tail; ++i) {        // `for box in invalid_box`, appended boxes included
            const CutBox b = work.get(i);
            if (!(b.b & kAlive)) continue;
            if (skip) {                         // this box slid under the iterator when its predecessor was removed
                skip = false;
                continue;
            }
            const int bx = b.a & 255u, by = (b.a >> 8) & 255u, bz = (b.a >> 16) & 255u, low = b.b & 255u, high = (b.b >> 8) & 255u;
            int flags[3], nf = 0;               // mdCreator.py:60-66
            if (bx > hi) flags[nf++] = 0;
            if (by > hi) flags[nf++] = 1;
            if (bz > hi) flags[nf++] = 2;
            const int f = flags[rng.below((uint32_t)nf)];           // random.choice, :68
            int s1[5], s2[5];                   // x, y, z, low, high of the two parts
            if (f == 0) {                       // :70-79
                if (bx <= lo) continue;
                const int r = 1 + (int)rng.below((uint32_t)bx);         // random.randint(1, x)
                if (r < lo || bx - r < lo) continue;
                s1[0] = r, s1[1] = by, s1[2] = bz, s1[3] = low, s1[4] = high;
                s2[0] = bx - r, s2[1] = by, s2[2] = bz, s2[3] = low, s2[4] = high;
            } else if (f == 1) {                // :80-89
                if (by < lo) continue;
                const int r = 1 + (int)rng.below((uint32_t)by);
                if (r < lo || by - r < lo) continue;
                s1[0] = bx, s1[1] = r, s1[2] = bz, s1[3] = low, s1[4] = high;
                s2[0] = bx, s2[1] = by - r, s2[2] = bz, s2[3] = low, s2[4] = high;
            } else {                            // :90-99
                if (bz < lo) continue;
                const int r = 1 + (int)rng.below((uint32_t)bz);
                if (r < lo || bz - r < lo) continue;
                s1[0] = bx, s1[1] = by, s1[2] = bz - r, s1[3] = low, s1[4] = high - r;
                s2[0] = bx, s2[1] = by, s2[2] = r, s2[3] = high - r, s2[4] = high;
            }
            work.set(i, CutBox{b.a, b.b & ~kAlive});   // invalid_box.remove(box)
            skip = true;
            for (int part = 0; part < 2; ++part) {
                const int *c = part ? s2 : s1;
                const bool ok = c[0] >= lo && c[0] <= hi && c[1] >= lo && c[1] <= hi && c[2] >= lo && c[2] <= hi;
                if (ok)
                    vals.set(nv++, (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24));
                else
                    work.set(tail++, CutBox{(uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16),
                                            (uint32_t)c[3] | ((uint32_t)c[4] << 8) | kAlive});
            }
        }
        int alive = 0;                          // compact the survivors (order kept) for the next pass
        for (int i = 0; i < tail; ++i) {
            const CutBox b = work.get(i);
            if (b.b & kAlive) {
                if (alive != i) work.set(alive, b);
                ++alive;
            }
        }
        tail = alive;
        if (tail == 0) break;
    }
    // depart_box (:137-138): stable sort by the height of the base (the top byte), insertion sort
    for (int a = 1; a < nv; ++a) {
        const uint32_t v = vals.get(a);
        int k = a - 1;
        while (k >= 0 && (vals.get(k) >> 24) > (v >> 24)) {
            vals.set(k + 1, vals.get(k));
            --k;
        }
        vals.set(k + 1, v);
    }
    return nv;
}

// ---- CPython's random.Random on MT19937 with the state words at a stride (1 everywhere today) -------------------------------
struct StridedMT {
    uint32_t *mt;
    size_t stride;
    int idx;
    __host__ __device__ uint32_t &w(int i) { return mt[(size_t)i * stride]; }
    __host__ __device__ void init_genrand(uint32_t s) {
        w(0) = s;
        for (int i = 1; i < 624; ++i) {
            const uint32_t p = w(i - 1);
            w(i) = 1812433253u * (p ^ (p >> 30)) + (uint32_t)i;
        }
        idx = 624;
    }
    __host__ __device__ void seed(uint64_t seed) {   // random.seed(int): init_by_array over the 32-bit digits of the seed
        const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        const int klen = key[1] ? 2 : 1;
        init_genrand(19650218u);
        int i = 1, j = 0;
        for (int k = 624; k; --k) {
            const uint32_t p = w(i - 1);
            w(i) = (w(i) ^ ((p ^ (p >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++i >= 624) {
                w(0) = w(623);
                i = 1;
            }
            if (++j >= klen) j = 0;
        }
        for (int k = 623; k; --k) {
            const uint32_t p = w(i - 1);
            w(i) = (w(i) ^ ((p ^ (p >> 30)) * 1566083941u)) - (uint32_t)i;
            if (++i >= 624) {
                w(0) = w(623);
                i = 1;
            }
        }
        w(0) = 0x80000000u;
        idx = 624;
    }
    __host__ __device__ uint32_t u32() {
        if (idx >= 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (w(k) & 0x80000000u) | (w(k + 1 < 624 ? k + 1 : 0) & 0x7fffffffu);
                w(k) = w(k + 397 < 624 ? k + 397 : k - 227) ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = w(idx++);
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    __host__ __device__ uint32_t below(uint32_t n) { return getrandbits_below(*this, n); }
};

// host-side storages (bpp_gen_cut2): plain arrays
struct ArrayWork {
    CutBox *base;
    CutBox get(int i) const { return base[i]; }
    void set(int i, CutBox v) { base[i] = v; }
};
struct ArrayVals {
    uint32_t *base;
    uint32_t get(int i) const { return base[i]; }
    void set(int i, uint32_t v) { base[i] = v; }
};

// ---- per-bin generator record in bpp_stream.mt (opaque to callers; bpp_stream_sizes gives the size): kMtRec words per
// bin, contiguous:
//   [0, 32)        header: [0] index of the next unused output of the CURRENT state (0..624), [1] which half is current,
//                  [2] set when the other half already holds the state that FOLLOWS the current one;
//   [32, 1280)     two halves of 624 raw MT19937 state words;
//   [1280, 1656)   the OUTPUTS as BYTES: the top eight bits of every tempered word -- all a draw ever looks at:
//                  random.choice / random.randint on sides <= 255 are getrandbits(k <= 8) = output >> (32 - k) -- of half 0
//                  (624 bytes), of half 1 (624 bytes), then a copy of the first 256 bytes of half 0, so that a run of outputs
//                  that starts in half 1 and continues in its successor (half 0) is contiguous in memory like one that
//                  starts in half 0.  (Round 3 kept the tempered WORDS: 2 784 words per record; a refill's readers now
//                  fetch a quarter of the bytes and hold a quarter of the registers.)
// The fast pipeline regenerates states in a separate, fully parallel kernel (pretwist), so a generator that runs off the
// end of its state just changes halves, and consuming outputs is reading memory: nothing is buffered between refills.
constexpr int kMtHalf = 624;
constexpr int kMtPos = 0, kMtPar = 1, kMtNextOk = 2;
constexpr int kMtRaw = 32, kMtOut8 = kMtRaw + 2 * kMtHalf, kMtMirrorLen = 256;
constexpr int kMtOut8Bytes = 2 * kMtHalf + kMtMirrorLen;       // 1504
constexpr int kMtRec = kMtOut8 + kMtOut8Bytes / 4;           // 1656 words = 6 624 bytes
static_assert(kMtOut8Bytes % 4 == 0 && (kMtRec * 4) % 16 == 0, "records stay 16-byte aligned");
__host__ __device__ inline uint8_t *mt_out8(uint32_t *rec) { return (uint8_t *)(rec + kMtOut8); }
constexpr int kCtrRec = 4;      // BPP_STREAM_RNG_COUNTER: words per bin -- the 64-bit stream id (+ 2 reserved)

__host__ __device__ inline int stream_work_entries(int W, int L, int H, int lo) { return W * L * H / (lo * lo * lo) + 8; }

__host__ __device__ inline uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// ======================================================================================================================
// Refill, fast pipeline: four kernels per bpp_stream_refill.
//   scan     one lane per bin: which bins have used up rows since the last refill (jobs, bucketed by the number of
//            sequences they need so that a wave's lanes finish together) and which ring rows will be rewritten;
//   pretwist one wave per job whose generator has no successor state yet: the next 624 state words and outputs,
//            coalesced, through LDS;
//   cut      one lane per job, single-wave workgroups.  The list walk of mdCreator.py:117-135 runs one VISIT (split
//            attempt) per iteration: the two rejection loops of a visit -- random.choice over the long sides and
//            random.randint for the cut position, each `getrandbits until below n` -- are evaluated on eight outputs at
//            once (first accepted one wins, outputs before it are consumed), so that a visit is a short dependency chain
//            instead of 3.4 dependent iterations; a lane that finds no accepted output among its eight carries on in
//            the next iteration.  A wave alone on its SIMD is bound by dependent-instruction latency (~10 cycles per
//            instruction in branchy code), hence: no divergent branches -- everything is computed, selects pick what
//            applies, stores that do not apply go to a dummy word -- and work that can be done side by side.  Outputs
//            come from a 64-word LDS ring per lane, topped up every eight iterations from words loaded eight iterations
//            earlier.  The pending boxes live in two LDS lists per lane (this pass / survivors for the next pass: no
//            shifting, no compaction).  Cut boxes go straight into the ring row, unsorted, with their base height as
//            sort key;
//   sort     one wave per rewritten row: stable counting sort by base height (depart_box, :137-138), key stripped,
//            terminator padding.
// Same draws in the same order as cut2_generate above (tests/test_stream_supply.py runs both against the oracle's
// generator and Python's random module).
// ======================================================================================================================
struct StreamWork {        // views into bpp_stream.work (see plan_stream)
    int32_t *hdr;          // [16]: jobs in bucket 0..2, rows rewritten
    int32_t *jobs;         // [3][E]: local bin ids needing 1 / 2 / >= 3 sequences
    int32_t *target;       // [E]: gen_next every bin is brought to (the scan's reading of episode + depth: the step
                           //      kernels may be running beside the refill, the kernels must agree on one value)
    int64_t *rows;         // [D * E]: bin | episode << 32 of every row rewritten by this refill
    uint32_t *spill;       // [2 * nsp][nslots]: pending boxes beyond the LDS lists (rare)
    uint32_t *twist;       // [cut waves][kTwistWords]: scratch of the twist a cut wave does itself (a job on its second lap; rare)
    int32_t cap, nsp, nslots, maxn;
    int32_t fb;            // bits per field of a pending / unsorted box (stream_field_bits)
    int32_t kmax, urgent;  // kmax > 0: a bin gets at most kmax sequences per refill unless that leaves it fewer than
                           // `urgent` rows from its current episode (then as many as it takes); 0: always all depth rows
    int32_t stage;         // stream_cut_rows_kernel: boxes a lane's staging column holds
};
constexpr int kRowHdr = 2;             // entries in front of a ring row's items (include/bpp_abi.h): item 1 of the NEXT row and
                                       // item 0 of the row after it -- the look-ahead a step needs, in the line it reads anyway
constexpr int kOutRing = 64;           // output BYTES a lane holds in LDS, its own contiguous run of kRingStride bytes:
                                       // position p of the job's output sequence at byte p % 64, the first kCand bytes
                                       // repeated behind byte 63 so that kCand consecutive outputs are consecutive bytes
                                       // wherever they start
constexpr int kCand = 8;               // outputs a rejection loop looks at per iteration
constexpr int kRingStride = 76;        // 64 + 8 + a dummy word; 19 dwords per lane: an odd stride, aligned dword accesses of
                                       // the 32 lanes of a group fall into 32 different banks
constexpr int kRingDummy = kOutRing + kCand;   // byte offset of the lane's dummy word
constexpr int kOutFetch = 48;          // output bytes loaded per top-up (twelve packed words)
constexpr int kTopUpEvery = 8;         // iterations between top-ups
constexpr int kTwistWords = 640;       // scratch of a wave-wide twist (624 used)
constexpr int kSortMaxT = 2048;        // longest row the sort kernel stages in LDS
constexpr int kScanThreads = 1024;

// LDS entries per pending list for sequences of at most maxn boxes.  Measured peaks: 10^3 bins 10 on average, above 16
// in 0.5 % of the sequences, 21 at most in 3000; 20^3 bins 68.  Longer lists continue in global memory (the wave then
// runs its general iteration).
__host__ inline int stream_pend_cap(int maxn) {
    int c = (maxn / 18 + 16 + 7) / 8 * 8;
    c = c < 16 ? 16 : (c > 80 ? 80 : c);
    return c < maxn ? c : maxn;
}
// A pending box is four fields of FB bits: x | y << FB | z << 2 FB | base height << 3 FB.  Bins whose sides are all below
// 16 (every 10^3-class bin) use FB = 4: a box is 16 bits and the LDS lists are arrays of half-words -- half the LDS of the
// round-3 kernel's lists, which together with the byte-sized outputs takes a cut wave from 31 KB of LDS to 11 KB (the step
// kernel's workgroups need 18.9 KB each and the two kernels run side by side).  FB = 8 for anything larger.
__host__ __device__ inline int stream_field_bits(int W, int L, int H) { return (W < 16 && L < 16 && H < 16) ? 4 : 8; }
// bytes of LDS of a cut wave: the two lists + the dummy entry, the rings
__host__ __device__ inline int stream_cut_lds_bytes(int cap, int fb) {
    return ((2 * cap + 1) * 64 * (fb == 4 ? 2 : 4) + 15) / 16 * 16 + 64 * kRingStride;
}

__global__ __launch_bounds__(kScanThreads) void stream_scan_kernel(bpp_stream s, StreamWork w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * [waves][4] counters
    int (*wave_cnt)[4] = (int (*)[4])smem, (*wave_base)[4] = wave_cnt + kScanThreads / 64;
    const int e = blockIdx.x * kScanThreads + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = s.num_envs, D = s.depth;
    int need = 0, g0 = 0;
    if (e < E) {
        g0 = s.gen_next[e];
        const int margin = g0 - s.state[e].episode;       // rows from the current episode on
        need = D - margin;
        if (w.kmax > 0) need = min(need, max(w.kmax, w.urgent - margin));
        need = need < 0 ? 0 : need;
        w.target[e] = g0 + need;
    }
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int bucket = need >= 3 ? 2 : need - 1;
    const int nr = need < D ? need : D;      // rows: only the last D sequences of a bin survive in the ring
    int incl = nr;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    uint64_t m[3];
    for (int b = 0; b < 3; ++b) m[b] = __ballot(bucket == b);
    if (lane < 3) wave_cnt[wave][lane] = __popcll(m[lane]);
    if (lane == 63) wave_cnt[wave][3] = incl;
    __syncthreads();
    if (threadIdx.x < 4) {                   // one atomic per workgroup and counter
        int tot = 0;
        for (int k = 0; k < kScanThreads / 64; ++k) {
            wave_base[k][threadIdx.x] = tot;
            tot += wave_cnt[k][threadIdx.x];
        }
        const int base = tot ? __hip_atomic_fetch_add(&w.hdr[threadIdx.x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0;
        for (int k = 0; k < kScanThreads / 64; ++k) wave_base[k][threadIdx.x] += base;
    }
    __syncthreads();
    if (bucket >= 0) w.jobs[(size_t)bucket * E + wave_base[wave][bucket] + __popcll(m[bucket] & below)] = e;
    const int first = g0 + need - nr, at = wave_base[wave][3] + incl - nr;
    for (int k = 0; k < nr; ++k) w.rows[at + k] = (int64_t)(uint32_t)e | ((int64_t)(first + k) << 32);
}

// The whole wave computes the state that follows half `par` of a record into the other half: raw words, output bytes
// and, for half 0, the mirror (tw: scratch of kTwistWords words).  Word k needs the OLD words k and k+1 and word k+397
// (old for k < 227, else the NEW word k-227); walking k in rounds of 64 consecutive words with all reads of a round before
// its writes gives every lane exactly those values (word 623 reads the new word 0, as the serial loop does).
// GLOBAL_TW: the scratch is global memory (the cut kernel, whose LDS is the pending lists and rings and nothing else):
// other lanes' words are then ordered by a workgroup-scope release / acquire around the wave barrier instead of the
// wavefront-scope one that suffices for LDS.
template <bool GLOBAL_TW>
__device__ __forceinline__ void twist_sync() {
    if constexpr (GLOBAL_TW) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        wave_sync();
    }
}
template <bool GLOBAL_TW>
__device__ __forceinline__ void stream_wave_twist(uint32_t *rec, uint32_t par, uint32_t *tw, int lane) {
    const uint32_t *src = rec + kMtRaw + par * kMtHalf;
    for (int k = lane; k < 624; k += 64) tw[k] = src[k];
    twist_sync<GLOBAL_TW>();
    for (int k0 = 0; k0 < 624; k0 += 64) {
        const int k = k0 + lane;
        uint32_t a = 0, b = 0, c = 0;
        if (k < 624) {
            a = tw[k];
            b = tw[k + 1 < 624 ? k + 1 : 0];
            c = tw[k + 397 < 624 ? k + 397 : k - 227];
        }
        twist_sync<GLOBAL_TW>();
        if (k < 624) {
            const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
            tw[k] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        twist_sync<GLOBAL_TW>();
    }
    const uint32_t to = par ^ 1u;
    uint8_t *out8 = mt_out8(rec);
    for (int k = lane; k < 624; k += 64) {
        const uint32_t y = tw[k];
        const uint8_t t = (uint8_t)(mt_temper(y) >> 24);
        rec[kMtRaw + to * kMtHalf + k] = y;
        out8[to * kMtHalf + k] = t;
        if (to == 0u && k < kMtMirrorLen) out8[2 * kMtHalf + k] = t;
    }
}

__global__ __launch_bounds__(256) void stream_pretwist_kernel(bpp_stream s, StreamWork w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [4][kTwistWords]
    uint32_t (*scratch)[kTwistWords] = (uint32_t (*)[kTwistWords])smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = s.num_envs;
    const int n3 = w.hdr[2], n2 = w.hdr[1], n1 = w.hdr[0];
    for (int q = blockIdx.x * 4 + wave; q < n3 + n2 + n1; q += gridDim.x * 4) {     // wave-uniform
        const int bucket = q < n3 ? 2 : (q < n3 + n2 ? 1 : 0);
        const int j = q - (bucket == 2 ? 0 : (bucket == 1 ? n3 : n3 + n2));
        uint32_t *rec = s.mt + (size_t)w.jobs[(size_t)bucket * E + j] * kMtRec;
        if (rec[kMtNextOk] == 0u) {
            const uint32_t par = rec[kMtPar];
            wave_sync();
            stream_wave_twist<false>(rec, par, scratch[wave], lane);
            if (lane == 0) rec[kMtNextOk] = 1u;
        }
        wave_sync();
    }
}

// The LDS part of a lane's two pending lists (+ one dummy entry): entry w of this lane at index w * 64 of an array of
// half-words (FB = 4) or words (FB = 8).
template <int FB, int STRIDE = 64>
struct ListCol {
    using T = typename std::conditional<FB == 4, uint16_t, uint32_t>::type;
    T *p;               // &array[lane]  (STRIDE 1: a lane that has the array to itself, stream_cut_rows_kernel's second attempt)
    __device__ __forceinline__ uint32_t get(uint32_t w) const { return (uint32_t)p[w * STRIDE]; }
    __device__ __forceinline__ void set(uint32_t w, uint32_t v) const { p[w * STRIDE] = (T)v; }
};

// the pending lists of one lane with their continuation in global memory: list r (0 / 1), entry i
template <int FB>
struct PendLists {
    ListCol<FB> lds;
    uint32_t *spill;    // &spill[slot]: entry k of list r beyond the LDS part at spill[(r * nsp + k) * nslots]
    int cap, nsp;
    size_t nslots;
    __device__ __forceinline__ uint32_t get(int r, int i) const {
        return i < cap ? lds.get((uint32_t)(r * cap + i)) : spill[(size_t)(r * nsp + i - cap) * nslots];
    }
    __device__ __forceinline__ void set(int r, int i, uint32_t v) const {
        if (i < cap) lds.set((uint32_t)(r * cap + i), v);
        else spill[(size_t)(r * nsp + i - cap) * nslots] = v;
    }
};

// State of one lane's list walk.  A pending box is x | y << FB | z << 2 FB | base height << 3 FB (its top is base + z).
struct CutLane {
    uint32_t box, v;        // box being visited; side being cut (valid in state 1)
    int st, f;              // 0: the next output chooses the side (random.choice), 1: it is the cut position (randint)
    int i, tail_a, tail_b;  // position in this pass's list, its length, length of the survivors' list
    int side;               // which LDS list is this pass's (0 / 1)
    int nv;                 // boxes cut so far
    uint32_t *row;
    int used;               // outputs consumed since the job started
};

// The kCand output bytes that follow position `from` of the lane's ring, as two packed words (byte j of the run in bits
// 8 j .. 8 j + 7 of lo for j < 4, of hi for j >= 4): three aligned dword reads (the repeated first bytes make the run
// contiguous across the end of the ring) and two byte alignments.
__device__ __forceinline__ void ring_run(const uint8_t *ringb, int from, uint32_t &lo, uint32_t &hi) {
    const uint32_t pos = (uint32_t)from & (kOutRing - 1);
    const uint32_t *pw = (const uint32_t *)(ringb + (pos & ~3u));
    const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2];
    lo = __builtin_amdgcn_alignbyte(w1, w0, pos & 3u);
    hi = __builtin_amdgcn_alignbyte(w2, w1, pos & 3u);
}

// `getrandbits(k) until below lim` on the (at most kCand) outputs the lane holds from position `from` of its ring:
// index of the first accepted output and its value; kCand when none is accepted.  k = bit_length(lim) <= 8: the draw is
// the top k bits of the output's byte.
__device__ __forceinline__ void first_below(const uint8_t *ringb, int from, uint32_t k, uint32_t lim, int &first, uint32_t &x) {
    uint32_t lo, hi;
    ring_run(ringb, from, lo, hi);
    first = kCand;
    x = 0;
#pragma unroll
    for (int j = kCand - 1; j >= 0; --j) {
        const uint32_t xj = (((j < 4 ? lo : hi) >> (8 * (j & 3))) & 255u) >> (8u - k);
        const bool a = xj < lim;
        first = a ? j : first;
        x = a ? xj : x;
    }
}

// ---- predicates as all-ones / zero words.  On this part a v_cndmask_b32 costs as much as five to nine plain vector
// instructions (tools/ubench.hip: 9.8 ns per wave against 1.1 - 1.8 ns), so the branch-free visit below selects with
// and / or / bit-field-insert on such masks instead.  All operands are small non-negative numbers (< 2^31).
__device__ __forceinline__ uint32_t m_lt(uint32_t a, uint32_t b) { return (uint32_t)((int32_t)(a - b) >> 31); }   // a < b
__device__ __forceinline__ uint32_t m_eq(uint32_t a, uint32_t b) { return m_lt(a ^ b, 1u); }
__device__ __forceinline__ uint32_t m_sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }       // m ? a : b

// `getrandbits(k) until below lim` on the kCand outputs that follow position `from` of the lane's ring: the first accepted
// output wins.  Every output becomes a key -- position << 8 | value when accepted, above 0xffff when not -- and the
// smallest key is the answer; the keys are independent of each other, which is what turns a visit's serial chain of
// draws into work done side by side.  Returns the key: position = key >> 8 (>= 256: none accepted), value = key & 255.
// The draw of output j is the top k bits of its byte: both packed words are shifted right by 8 - k once, after which
// the draw sits in bits 8 j .. 8 j + k - 1 -- one bit-field extract per output (what is left of the neighbour byte above
// it lies outside the field).
__device__ __forceinline__ uint32_t first_below_key(const uint8_t *ringb, int from, uint32_t k, uint32_t lim) {
    uint32_t lo, hi;
    ring_run(ringb, from, lo, hi);
    lo >>= 8u - k;
    hi >>= 8u - k;
    const uint32_t lim1 = lim - 1u;
    uint32_t key[kCand];
#pragma unroll
    for (int j = 0; j < kCand; ++j) {
        const uint32_t xj = __builtin_amdgcn_ubfe(j < 4 ? lo : hi, 8u * (uint32_t)(j & 3), k);
        key[j] = (((lim1 - xj) >> 31) << 16) | (xj | ((uint32_t)j << 8));
    }
    static_assert(kCand == 8, "min tree below is written for eight candidates");
    return min(min(min(key[0], key[1]), min(key[2], key[3])), min(min(key[4], key[5]), min(key[6], key[7])));
}
__device__ __forceinline__ uint32_t bit_length(uint32_t v) { return 32u - (uint32_t)__clz((int)v); }   // v > 0

// One visit for a lane whose lists are certain to stay inside LDS (tail_a + 2 <= cap, tail_b + 1 <= cap): the statement
// of cut_visit_general below without divergent branches and without selects on condition codes.  `act` is the lane's
// all-ones / zero activity mask; returns the mask "sequence complete".
template <int FB>
__device__ __forceinline__ uint32_t cut_visit_lds(CutLane &c, const ListCol<FB> col, int cap, const uint8_t *ringb, int filled, uint32_t act,
                                                  uint32_t lo, uint32_t hi) {
    constexpr uint32_t FM = (1u << FB) - 1u;
    const uint32_t dummy = 2u * (uint32_t)cap;
    const uint32_t abase = (uint32_t)cap & (0u - (uint32_t)c.side), bbase = (uint32_t)cap - abase;
    const uint32_t box = c.box;
    const uint32_t bx = box & FM, by = (box >> FB) & FM, bz = (box >> (2 * FB)) & FM;
    const uint32_t mfx = m_lt(hi, bx), mfy = m_lt(hi, by), mfz = m_lt(hi, bz);          // :60-66
    const uint32_t nf = 0u - (mfx + mfy + mfz);
    const uint32_t monly = m_eq(nf, 1u);
    const uint32_t mst0 = (uint32_t)c.st - 1u;                                          // st is 0 or 1
    // random.choice(flags) (:68) -- getrandbits(1) for one long side, getrandbits(2) for two or three --, or
    // random.randint(1, v) (:73 / :83 / :93) for a lane that chose its side earlier
    const uint32_t key1 = first_below_key(ringb, c.used, m_sel(mst0, 2u + monly, bit_length(c.v)), m_sel(mst0, nf, c.v));
    const uint32_t have1 = (uint32_t)min(filled - c.used, kCand);
    const uint32_t first1 = key1 >> 8, x1 = key1 & 255u;
    const uint32_t mfound1 = m_lt(first1, have1) & act;
    c.used += (int)(m_sel(mfound1, first1 + 1u, have1) & act);
    const uint32_t mchoose = mfound1 & mst0;
    const uint32_t f0 = (2u + mfy) & ~mfx;                                              // first long side
    const uint32_t f1 = 2u + (mfx & mfy);                                               // second long side
    const uint32_t fnew = m_sel(m_lt(x1, 1u), f0, m_sel(m_eq(x1, 1u), f1, 2u));
    const uint32_t f = m_sel(mchoose, fnew, (uint32_t)c.f);
    const uint32_t v = m_sel(mchoose, (box >> ((uint32_t)FB * fnew)) & FM, c.v);
    // the cut position for a side chosen just now
    const uint32_t key2 = first_below_key(ringb, c.used, bit_length(v), v);
    const uint32_t have2 = (uint32_t)min(filled - c.used, kCand);
    const uint32_t first2 = key2 >> 8, x2 = key2 & 255u;
    const uint32_t mfound2 = m_lt(first2, have2) & mchoose;
    c.used += (int)(m_sel(mfound2, first2 + 1u, have2) & mchoose);
    const uint32_t mfin = (mfound1 & ~mst0) | mfound2;                                  // the visit is decided
    const uint32_t r = m_sel(mfound2, x2, x1) + 1u;
    const uint32_t mgood = ~(m_lt(r, lo) | m_lt(v - r, lo));                            // :74, :84, :94
    const uint32_t msplit = mfin & mgood, mfail = mfin & ~mgood;
    const uint32_t mf2 = m_eq(f, 2u);
    const uint32_t sh = (uint32_t)FB * f, p1 = m_sel(mf2, v - r, r), p2 = v - p1;
    const uint32_t rest = box & ~(FM << sh);
    const uint32_t c1 = rest | (p1 << sh);
    const uint32_t c2 = (rest | (p2 << sh)) + ((p1 << (3 * FB)) & mf2);                 // :97-98: the upper part starts at high - r
    const uint32_t me1 = msplit & monly & ~m_lt(hi, p1), me2 = msplit & monly & ~m_lt(hi, p2);   // is_valid, :110-115
    const uint32_t mq1 = msplit & ~me1, mq2 = msplit & ~me2;
    uint32_t nv = (uint32_t)c.nv, tail_a = (uint32_t)c.tail_a, tail_b = (uint32_t)c.tail_b, i = (uint32_t)c.i;
    if (me1) c.row[nv] = c1;
    nv -= me1;
    if (me2) c.row[nv] = c2;
    nv -= me2;
    col.set(m_sel(mfail, bbase + tail_b, dummy), box);            // stays in invalid_box for the next pass
    tail_b -= mfail;
    col.set(m_sel(mq1, abase + tail_a, dummy), c1);               // appended: visited later in this pass
    tail_a -= mq1;
    col.set(m_sel(mq2, abase + tail_a, dummy), c2);
    tail_a -= mq2;
    c.st = (int)(((uint32_t)c.st | (mchoose & 1u)) & ~mfin);
    c.f = (int)f;
    c.v = v;
    i -= mfin;
    // the two entries after the visited box (the first is skipped after a split, :124) and the head of the survivors
    const uint32_t n1 = col.get(abase + min(i, (uint32_t)cap - 1u)), n2 = col.get(abase + min(i + 1u, (uint32_t)cap - 1u));
    const uint32_t mskip = msplit & m_lt(i, tail_a);
    col.set(m_sel(mskip, bbase + tail_b, dummy), n1);
    tail_b -= mskip;
    i -= mskip;
    const uint32_t b0 = col.get(bbase);
    const uint32_t mpass = mfin & ~m_lt(i, tail_a);               // end of the `for`: next pass over the survivors, or done
    const uint32_t mdone = mpass & m_eq(tail_b, 0u);
    c.box = m_sel(mfin, m_sel(mpass, b0, m_sel(mskip, n2, n1)), box);
    c.side ^= (int)(mpass & 1u);
    c.tail_a = (int)m_sel(mpass, tail_b, tail_a);
    c.tail_b = (int)(tail_b & ~mpass);
    c.i = (int)(i & ~mpass);
    c.nv = (int)nv;
    return mdone;
}

// One visit (mdCreator.py:59-100 benchmark_split and its bookkeeping in gen_benchmark :120-130), or the part of it the
// lane has outputs for, in plain form: any list length (entries beyond the LDS part live in global memory).  `filled` =
// outputs put into the ring so far.  Returns true when the sequence is complete.  This is the statement
// cut_visit_lds above is checked against: a wave runs it whenever one of its lanes' lists may leave LDS.
// (`f == 0 ? v <= lo : v < lo`, :71 / :81 / :91, cannot hold: the side was chosen because it exceeds hi, and bpp_stream
// requires hi >= 2 lo - 1 >= lo; cut2_generate keeps the test.)
template <int FB>
__device__ __forceinline__ bool cut_visit_general(CutLane &c, const PendLists<FB> &pend, const uint8_t *ringb, int filled, uint32_t lo,
                                                  uint32_t hi) {
    constexpr uint32_t FM = (1u << FB) - 1u;
    const uint32_t bx = c.box & FM, by = (c.box >> FB) & FM, bz = (c.box >> (2 * FB)) & FM;
    const bool fx = bx > hi, fy = by > hi, fz = bz > hi;                    // :60-66
    const uint32_t nf = (uint32_t)fx + (uint32_t)fy + (uint32_t)fz;
    const bool st0 = c.st == 0;
    // first rejection loop of the iteration: random.choice(flags) (:68) = flags[getrandbits(bit_length(nf)) until < nf],
    // or, for a lane that chose its side earlier, random.randint(1, v) (:73 / :83 / :93) = 1 + (getrandbits(bit_length(v)) until < v)
    int first1, first2;
    uint32_t x1, x2;
    first_below(ringb, c.used, st0 ? (nf == 1u ? 1u : 2u) : bit_length(c.v), st0 ? nf : c.v, first1, x1);
    const int have1 = min(filled - c.used, kCand);
    const bool found1 = first1 < have1;
    c.used += found1 ? first1 + 1 : have1;
    const bool choose = found1 && st0;
    if (choose) {
        c.f = x1 == 0u ? (fx ? 0 : (fy ? 1 : 2)) : (x1 == 1u ? ((fx && fy) ? 1 : 2) : 2);
        c.v = c.f == 0 ? bx : (c.f == 1 ? by : bz);
        c.st = 1;
    }
    // second one: the cut position for a side chosen just now
    first_below(ringb, c.used, bit_length(c.v), c.v, first2, x2);
    const int have2 = min(filled - c.used, kCand);
    const bool found2 = choose && first2 < have2;
    if (choose) c.used += found2 ? first2 + 1 : have2;
    if (!((found1 && !st0) || found2)) return false;                         // the visit is not decided yet
    const uint32_t r = (found2 ? x2 : x1) + 1u;
    const bool split = r >= lo && c.v - r >= lo;                              // :74, :84, :94
    c.st = 0;
    if (!split) {
        pend.set(c.side ^ 1, c.tail_b++, c.box);                              // stays in invalid_box for the next pass
    } else {
        const uint32_t sh = (uint32_t)FB * (uint32_t)c.f, p1 = c.f == 2 ? c.v - r : r, p2 = c.v - p1;
        const uint32_t rest = c.box & ~(FM << sh);
        const uint32_t c1 = rest | (p1 << sh);
        const uint32_t c2 = (rest | (p2 << sh)) + (c.f == 2 ? p1 << (3 * FB) : 0u);   // :97-98: the upper part starts at high - r
        // is_valid (:110-115): the untouched sides are within bounds iff the cut side was the only long one
        if (nf == 1u && p1 <= hi) c.row[c.nv++] = c1;
        else pend.set(c.side, c.tail_a++, c1);                                // appended: visited later in this pass
        if (nf == 1u && p2 <= hi) c.row[c.nv++] = c2;
        else pend.set(c.side, c.tail_a++, c2);
    }
    ++c.i;
    if (split && c.i < c.tail_a) {                // the removal slid the next box under the iterator: not visited in this pass
        pend.set(c.side ^ 1, c.tail_b++, pend.get(c.side, c.i));
        ++c.i;
    }
    if (c.i < c.tail_a) {
        c.box = pend.get(c.side, c.i);
        return false;
    }
    const bool finished = c.tail_b == 0;          // end of the `for`: next pass over the survivors, or done
    c.side ^= 1;
    c.tail_a = c.tail_b;
    c.tail_b = 0;
    c.i = 0;
    if (!finished) c.box = pend.get(c.side, 0);
    return finished;
}

template <int FB>
__global__ __launch_bounds__(64) void stream_cut_kernel(bpp_stream s, StreamWork w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int E = s.num_envs, T = s.pool_len, D = s.depth, cap = w.cap;
    const uint32_t lo = (uint32_t)s.bound_lo, hi = (uint32_t)s.bound_hi;
    // This wave is one long dependency chain that uses a fraction of its SIMD's issue slots; beside the step kernel (eight
    // waves per SIMD that want every slot) it must win the arbitration whenever it can issue, or the refill falls behind
    // the lock-steps it runs beside.
    __builtin_amdgcn_s_setprio(3);
    // wave -> (bucket, position): the longest jobs are dispatched first
    const int n3 = w.hdr[2], n2 = w.hdr[1], n1 = w.hdr[0];
    const int w3 = (n3 + 63) >> 6, w2 = (n2 + 63) >> 6, w1 = (n1 + 63) >> 6;
    int b = blockIdx.x, n, bucket;
    if (b < w3) n = n3, bucket = 2;
    else if (b < w3 + w2) b -= w3, n = n2, bucket = 1;
    else if (b < w3 + w2 + w1) b -= w3 + w2, n = n1, bucket = 0;
    else return;
    const int j = b * 64 + lane;
    const bool job = j < n;
    const int e = job ? w.jobs[(size_t)bucket * E + j] : 0;
    uint32_t *rec = s.mt + (size_t)e * kMtRec;
    const ListCol<FB> col{(typename ListCol<FB>::T *)smem + lane};      // entry w of this lane at index w * 64
    // outputs: byte p % 64 of this lane's run holds output p (counted from the job's start)
    uint8_t *ringb = smem + ((size_t)(2 * cap + 1) * 64 * sizeof(typename ListCol<FB>::T) + 15) / 16 * 16 + (size_t)lane * kRingStride;
    uint32_t *tw = w.twist + (size_t)blockIdx.x * kTwistWords;        // twist scratch of the wave (global: the path is rare)
    const PendLists<FB> pend{col, w.spill + (size_t)blockIdx.x * 64 + lane, cap, w.nsp, (size_t)w.nslots};

    int g = 0, need = 0, base = 0;                                   // base: index in the current state of the job's first output
    uint32_t par = 0, next_ok = 0;
    if (job) {
        base = (int)rec[kMtPos];
        par = rec[kMtPar];
        next_ok = rec[kMtNextOk];
        g = s.gen_next[e];
        need = w.target[e] - g;
    }
    bool active = job && need > 0;
    const bool ran = active;
    const uint32_t whole = (uint32_t)s.W | ((uint32_t)s.L << FB) | ((uint32_t)s.H << (2 * FB));
    CutLane c{whole, 1u, 0, 0, 0, 1, 0, 0, 0, (uint32_t *)s.ring + ((size_t)(active ? g % D : 0) * E + e) * T + kRowHdr, 0};
    if (active) col.set(0u, whole);
    int filled = 0;                                                  // outputs put into the ring since the job started (a multiple of 4)

    // The next kOutFetch output bytes after `filled`, loaded one top-up ahead as twelve packed words.  Outputs of the
    // current state and of its successor are contiguous in the record (mirror); the successor is there (pretwist kernel)
    // -- except for a job on its second lap, for which the wave makes it now.
    uint32_t pre[kOutFetch / 4];
    auto fetch = [&]() {
        uint64_t m = __ballot(active && base + filled + kOutFetch > kMtHalf && !next_ok);
        while (m) {                                   // wave-uniform, rare: one bin at a time, all lanes help
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const int el = (int)__builtin_amdgcn_readlane(e, l);
            const uint32_t pl = __builtin_amdgcn_readlane(par, l);
            twist_sync<true>();
            stream_wave_twist<true>(s.mt + (size_t)el * kMtRec, pl, tw, lane);
            twist_sync<true>();                          // the new outputs are read by lane l's fetch right below
            if (lane == l) next_ok = 1u;
        }
        // in-bounds for every lane, read or not: thirteen aligned words around the 48 bytes, shifted into place
        const uint8_t *p8 = mt_out8(rec) + par * kMtHalf + base + filled;
        const uint32_t off = (uint32_t)((uintptr_t)p8 & 3u);
        const uint32_t *pw = (const uint32_t *)(p8 - off);
        uint32_t raw[kOutFetch / 4 + 1];
#pragma unroll
        for (int q = 0; q <= kOutFetch / 4; ++q) raw[q] = pw[q];
#pragma unroll
        for (int q = 0; q < kOutFetch / 4; ++q) pre[q] = __builtin_amdgcn_alignbyte(raw[q + 1], raw[q], off);
    };
    auto top_up = [&]() {
        const int room = kOutRing - (filled - c.used);
        const int put = active ? (min(kOutFetch, room) & ~3) : 0;     // whole words: `filled` stays a multiple of 4
#pragma unroll
        for (int q = 0; q < kOutFetch / 4; ++q) {
            const uint32_t slot = (uint32_t)(filled + 4 * q) & (kOutRing - 1), mput = m_lt((uint32_t)(4 * q), (uint32_t)put);
            *(uint32_t *)(ringb + m_sel(mput, slot, (uint32_t)kRingDummy)) = pre[q];                        // (else: the dummy word)
            *(uint32_t *)(ringb + m_sel(mput & m_lt(slot, (uint32_t)kCand), slot + kOutRing, (uint32_t)kRingDummy)) = pre[q];   // repeated first bytes
        }
        filled += put;
        if (active && base + c.used >= kMtHalf) {     // the lane now draws from the successor: it becomes the current state
            base -= kMtHalf;
            par ^= 1u;
            next_ok = 0u;
        }
    };
    fetch();
    int it = 0;
    while (__ballot(active)) {
        if ((it & (kTopUpEvery - 1)) == 0) {          // wave-uniform
            top_up();
            fetch();
        }
        bool finished;
        if (__ballot(active && (c.tail_a + 2 > cap || c.tail_b + 1 > cap))) {   // wave-uniform: a list may leave LDS
            finished = active ? cut_visit_general<FB>(c, pend, ringb, filled, lo, hi) : false;
        } else {
            finished = cut_visit_lds<FB>(c, col, cap, ringb, filled, active ? ~0u : 0u, lo, hi) != 0u;
        }
        if (finished) {
            c.row[T - 1 - kRowHdr] = (uint32_t)c.nv;  // (the row's last entry) length for the sort kernel, which restores the terminator
            ++g;
            if (--need > 0) {
                c.row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T + kRowHdr;
                pend.set(c.side, 0, whole);
                c.tail_a = 1;
                c.nv = 0;
                c.box = whole;
            } else {
                active = false;
            }
        }
        ++it;
    }
    if (ran) {                                        // outputs in the ring that were not used are simply read again next time
        int pos = base + c.used;
        if (pos >= kMtHalf) {
            pos -= kMtHalf;
            par ^= 1u;
            next_ok = 0u;
        }
        rec[kMtPos] = (uint32_t)pos;
        rec[kMtPar] = par;
        rec[kMtNextOk] = next_ok;
        s.gen_next[e] = g;
    }
}

// One wave per rewritten row: stable counting sort of the cut boxes by the height of their base (depart_box, :137-138),
// sort key stripped, the rest of the row padded with the terminator.  Rows of at most 64 boxes (every 10^3 row: 47 at
// most) never touch LDS: a lane holds one box, its place is the number of boxes with a lower base plus the number of
// equal ones in lower lanes, counted with one ballot per distinct base height; the next row's boxes are loaded before
// the current row is ranked.  Longer rows are staged in LDS and ranked chunk by chunk.
// ======================================================================================================================
// BPP_STREAM_RNG_COUNTER, fast pipeline: scan / cut / sort (no pretwist: there is no state to regenerate).  The cut kernel
// is the list walk of stream_cut_kernel without everything the Mersenne Twister forced on it: a draw is a hash of (key,
// draw index) -- no ring of outputs, no top-ups, no fetches --, a rejection happens once in 2^32 / lim draws -- no
// eight-candidate min-trees, a visit is always decided in the iteration that starts it.  Same lists (ListCol / PendLists),
// same unsorted row format, same sort kernel.
// ======================================================================================================================
struct CtrLane {
    uint32_t box;
    int i, tail_a, tail_b;  // position in this pass's list, its length, length of the survivors' list
    int side;               // which LDS list is this pass's (0 / 1)
    int nv;                 // boxes cut so far
    uint32_t *row;
    CtrRng rng;
};

// One visit for a lane whose lists are certain to stay inside LDS: mdCreator.py:59-100 + :120-130 without divergent
// branches (the one that exists -- a rejected draw -- is taken by nobody in 2^32 / lim - 1 of 2^32 / lim cases).
// Where a visit's finished boxes go: straight into the ring row, unsorted (the sort kernel ranks them later) ...
struct EmitToRow {
    uint32_t *row;
    __device__ __forceinline__ void operator()(uint32_t m, uint32_t nv, uint32_t v) const {
        if (m) row[nv] = v;
    }
};
// ... or into the lane's staging column in LDS (stream_cut_rows_kernel, which ranks them itself); entry `dummy` takes the
// stores that do not apply;
// there it is also counted (level `nolevel` for the stores that do not apply).  Level counters of a lane: two 16-bit counters
// per word -- a row has fewer than 2 048 boxes --, level l in half l & 1 of word (l >> 1) * 64.
__device__ __forceinline__ uint32_t level_count(uint32_t *cnt, uint32_t level) {      // returns the level's count before
    const uint32_t sh = (level & 1u) * 16u;
    return (atomicAdd(&cnt[(level >> 1) * 64], 1u << sh) >> sh) & 0xffffu;
}
template <class Col, int FB>
struct EmitToStage {
    Col st;
    uint32_t *cnt;
    uint32_t dummy, nolevel;
    __device__ __forceinline__ void operator()(uint32_t m, uint32_t nv, uint32_t v) const {
        st.set(m_sel(m, nv, dummy), v);
        const uint32_t level = m_sel(m, v >> (3 * FB), nolevel);
        atomicAdd(&cnt[(level >> 1) * 64], 1u << ((level & 1u) * 16u));
    }
};

template <int FB, class Col, class Emit>
__device__ __forceinline__ uint32_t cut_visit_ctr(CtrLane &c, const Col col, int cap, uint32_t act, uint32_t lo, uint32_t hi, const Emit emit) {
    constexpr uint32_t FM = (1u << FB) - 1u;
    const uint32_t dummy = 2u * (uint32_t)cap;
    const uint32_t abase = (uint32_t)cap & (0u - (uint32_t)c.side), bbase = (uint32_t)cap - abase;
    const uint32_t box = c.box;
    const uint32_t bx = box & FM, by = (box >> FB) & FM, bz = (box >> (2 * FB)) & FM;
    const uint32_t mfx = m_lt(hi, bx), mfy = m_lt(hi, by), mfz = m_lt(hi, bz);          // :60-66
    const uint32_t nf = max(0u - (mfx + mfy + mfz), 1u);                                // (1 for a lane without a box: no draw from 0)
    const uint32_t monly = m_eq(nf, 1u);
    const uint32_t x1 = c.rng.below(nf);                                                // random.choice(flags), :68
    const uint32_t f0 = (2u + mfy) & ~mfx;                                              // first long side
    const uint32_t f1 = 2u + (mfx & mfy);                                               // second long side
    const uint32_t f = m_sel(m_lt(x1, 1u), f0, m_sel(m_eq(x1, 1u), f1, 2u));
    const uint32_t v = max((box >> ((uint32_t)FB * f)) & FM, 1u);
    const uint32_t r = c.rng.below(v) + 1u;                                             // random.randint(1, v), :73 / :83 / :93
    const uint32_t mgood = ~(m_lt(r, lo) | m_lt(v - r, lo));                            // :74, :84, :94
    const uint32_t msplit = act & mgood, mfail = act & ~mgood;
    const uint32_t mf2 = m_eq(f, 2u);
    const uint32_t sh = (uint32_t)FB * f, p1 = m_sel(mf2, v - r, r), p2 = v - p1;
    const uint32_t rest = box & ~(FM << sh);
    const uint32_t c1 = rest | (p1 << sh);
    const uint32_t c2 = (rest | (p2 << sh)) + ((p1 << (3 * FB)) & mf2);                 // :97-98: the upper part starts at high - r
    const uint32_t me1 = msplit & monly & ~m_lt(hi, p1), me2 = msplit & monly & ~m_lt(hi, p2);   // is_valid, :110-115
    const uint32_t mq1 = msplit & ~me1, mq2 = msplit & ~me2;
    uint32_t nv = (uint32_t)c.nv, tail_a = (uint32_t)c.tail_a, tail_b = (uint32_t)c.tail_b, i = (uint32_t)c.i;
    emit(me1, nv, c1);
    nv -= me1;
    emit(me2, nv, c2);
    nv -= me2;
    col.set(m_sel(mfail, bbase + tail_b, dummy), box);            // stays in invalid_box for the next pass
    tail_b -= mfail;
    col.set(m_sel(mq1, abase + tail_a, dummy), c1);               // appended: visited later in this pass
    tail_a -= mq1;
    col.set(m_sel(mq2, abase + tail_a, dummy), c2);
    tail_a -= mq2;
    i -= act;
    // the two entries after the visited box (the first is skipped after a split, :124) and the head of the survivors
    const uint32_t n1 = col.get(abase + min(i, (uint32_t)cap - 1u)), n2 = col.get(abase + min(i + 1u, (uint32_t)cap - 1u));
    const uint32_t mskip = msplit & m_lt(i, tail_a);
    col.set(m_sel(mskip, bbase + tail_b, dummy), n1);
    tail_b -= mskip;
    i -= mskip;
    const uint32_t b0 = col.get(bbase);
    const uint32_t mpass = act & ~m_lt(i, tail_a);                // end of the `for`: next pass over the survivors, or done
    const uint32_t mdone = mpass & m_eq(tail_b, 0u);
    c.box = m_sel(act, m_sel(mpass, b0, m_sel(mskip, n2, n1)), box);
    c.side ^= (int)(mpass & 1u);
    c.tail_a = (int)m_sel(mpass, tail_b, tail_a);
    c.tail_b = (int)(tail_b & ~mpass);
    c.i = (int)(i & ~mpass);
    c.nv = (int)nv;
    return mdone;
}

// The same visit in plain form for lists of any length (entries beyond the LDS part live in global memory): the
// statement cut_visit_ctr is checked against; a wave runs it whenever one of its lanes' lists may leave LDS.
template <int FB>
__device__ __forceinline__ bool cut_visit_ctr_general(CtrLane &c, const PendLists<FB> &pend, uint32_t lo, uint32_t hi) {
    constexpr uint32_t FM = (1u << FB) - 1u;
    const uint32_t bx = c.box & FM, by = (c.box >> FB) & FM, bz = (c.box >> (2 * FB)) & FM;
    const bool fx = bx > hi, fy = by > hi, fz = bz > hi;                    // :60-66
    const uint32_t nf = (uint32_t)fx + (uint32_t)fy + (uint32_t)fz;
    const uint32_t x1 = c.rng.below(nf);                                    // random.choice(flags), :68
    const int f = x1 == 0u ? (fx ? 0 : (fy ? 1 : 2)) : (x1 == 1u ? ((fx && fy) ? 1 : 2) : 2);
    const uint32_t v = f == 0 ? bx : (f == 1 ? by : bz);
    const uint32_t r = c.rng.below(v) + 1u;                                 // random.randint(1, v)
    const bool split = r >= lo && v - r >= lo;                              // :74, :84, :94
    if (!split) {
        pend.set(c.side ^ 1, c.tail_b++, c.box);                            // stays in invalid_box for the next pass
    } else {
        const uint32_t sh = (uint32_t)FB * (uint32_t)f, p1 = f == 2 ? v - r : r, p2 = v - p1;
        const uint32_t rest = c.box & ~(FM << sh);
        const uint32_t c1 = rest | (p1 << sh);
        const uint32_t c2 = (rest | (p2 << sh)) + (f == 2 ? p1 << (3 * FB) : 0u);   // :97-98
        if (nf == 1u && p1 <= hi) c.row[c.nv++] = c1;                       // is_valid (:110-115)
        else pend.set(c.side, c.tail_a++, c1);
        if (nf == 1u && p2 <= hi) c.row[c.nv++] = c2;
        else pend.set(c.side, c.tail_a++, c2);
    }
    ++c.i;
    if (split && c.i < c.tail_a) {                // the removal slid the next box under the iterator: not visited in this pass
        pend.set(c.side ^ 1, c.tail_b++, pend.get(c.side, c.i));
        ++c.i;
    }
    if (c.i < c.tail_a) {
        c.box = pend.get(c.side, c.i);
        return false;
    }
    const bool finished = c.tail_b == 0;          // end of the `for`: next pass over the survivors, or done
    c.side ^= 1;
    c.tail_a = c.tail_b;
    c.tail_b = 0;
    c.i = 0;
    if (!finished) c.box = pend.get(c.side, 0);
    return finished;
}

template <int FB>
__global__ __launch_bounds__(64) void stream_cut_ctr_kernel(bpp_stream s, StreamWork w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int E = s.num_envs, T = s.pool_len, D = s.depth, cap = w.cap;
    const uint32_t lo = (uint32_t)s.bound_lo, hi = (uint32_t)s.bound_hi;
    // wave -> (bucket, position): the longest jobs are dispatched first
    const int n3 = w.hdr[2], n2 = w.hdr[1], n1 = w.hdr[0];
    const int w3 = (n3 + 63) >> 6, w2 = (n2 + 63) >> 6, w1 = (n1 + 63) >> 6;
    int b = blockIdx.x, n, bucket;
    if (b < w3) n = n3, bucket = 2;
    else if (b < w3 + w2) b -= w3, n = n2, bucket = 1;
    else if (b < w3 + w2 + w1) b -= w3 + w2, n = n1, bucket = 0;
    else return;
    const int j = b * 64 + lane;
    const bool job = j < n;
    const int e = job ? w.jobs[(size_t)bucket * E + j] : 0;
    const ListCol<FB> col{(typename ListCol<FB>::T *)smem + lane};
    const PendLists<FB> pend{col, w.spill + (size_t)blockIdx.x * 64 + lane, cap, w.nsp, (size_t)w.nslots};
    int g = 0, need = 0;
    uint64_t sid = 0;
    if (job) {
        const uint32_t *rc = s.mt + (size_t)e * kCtrRec;
        sid = (uint64_t)rc[0] | ((uint64_t)rc[1] << 32);
        g = s.gen_next[e];
        need = w.target[e] - g;
    }
    bool active = job && need > 0;
    const bool ran = active;
    const uint32_t whole = (uint32_t)s.W | ((uint32_t)s.L << FB) | ((uint32_t)s.H << (2 * FB));
    CtrLane c{whole, 0, 1, 0, 0, 0, (uint32_t *)s.ring + ((size_t)(active ? g % D : 0) * E + e) * T + kRowHdr, CtrRng::key(s.seed0, sid, (uint32_t)g)};
    if (active) col.set(0u, whole);
    while (__ballot(active)) {
        bool finished;
        if (__ballot(active && (c.tail_a + 2 > cap || c.tail_b + 1 > cap))) {   // wave-uniform: a list may leave LDS
            finished = active ? cut_visit_ctr_general<FB>(c, pend, lo, hi) : false;
        } else {
            finished = cut_visit_ctr<FB>(c, col, cap, active ? ~0u : 0u, lo, hi, EmitToRow{c.row}) != 0u;
        }
        if (finished) {
            c.row[T - 1 - kRowHdr] = (uint32_t)c.nv;  // (the row's last entry) length for the sort kernel, which restores the terminator
            ++g;
            if (--need > 0) {
                c.row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T + kRowHdr;
                pend.set(c.side, 0, whole);
                c.tail_a = 1;
                c.nv = 0;
                c.box = whole;
                c.rng = CtrRng::key(s.seed0, sid, (uint32_t)g);
            } else {
                active = false;
            }
        }
    }
    if (ran) s.gen_next[e] = g;
}

// An unsorted box as the cut kernel leaves it (fields of w.fb bits) -> x | y << 8 | z << 16 | base height << 24.
__device__ __forceinline__ uint32_t cut_box_bytes(uint32_t v, int fb) {
    return fb == 8 ? v : ((v & 15u) | ((v & 0xf0u) << 4) | ((v & 0xf00u) << 8) | ((v & 0xf000u) << 12));
}

// ======================================================================================================================
// BPP_STREAM_RNG_COUNTER, rows pipeline (round 5): scan / cut_rows.  The counter generator keys every SEQUENCE separately
// (seed0, stream id, episode), so the unit of work is a ring ROW, not a bin: one lane per row the scan listed, and the
// lane that cut a sequence also ranks it -- the boxes wait in an LDS column of the lane instead of the row, a counting
// sort by base height (depart_box, mdCreator.py:137-138: stable) over per-lane level counters in LDS puts them into the
// row in their final order, key stripped, and leaves the look-ahead copies of the first two items in the two rows before
// (what the sort kernel did from a cold read of the row, a hundred microseconds later).  The terminator padding behind the
// items is written by the whole wave, one row after the other, in full lines.
// A lane whose lists or whose staging column would overflow (capacities chosen so that it happens to one sequence in
// thousands) stops, and is served again once the others are done: alone, with the whole wave's LDS as its lists and
// staging area (entries at stride 1), which holds any sequence the geometry allows.
// ======================================================================================================================
__host__ __device__ inline int stream_rows_stage_cap(int W, int L, int H, int lo, int hi, int maxn) {
    // twice the typical length of a sequence (bin volume over the volume of a box of mean sides), at least 16, a multiple of 8
    const int mid2 = lo + hi;                   // 2 * mean side
    const long typical = (long)W * L * H * 8 / ((long)mid2 * mid2 * mid2);
    long c = (2 * typical + 7) / 8 * 8;
    c = c < 16 ? 16 : c;
    return (int)(c < maxn ? c : maxn);
}
// LDS entries of the lists and staging columns of a wave; the lane that is served alone needs 3 (maxn + 2) + 2 of them
__host__ __device__ inline size_t stream_rows_lds_entries(int cap, int stage) { return ((size_t)(2 * cap + 1) + (size_t)(stage + 1)) * 64; }
__host__ __device__ inline size_t stream_rows_lds_bytes(int cap, int stage, int fb, int H) {
    return (stream_rows_lds_entries(cap, stage) * (fb == 4 ? 2 : 4) + 15) / 16 * 16 + (size_t)((H + 2) / 2) * 64 * 4;
}

// Sequence `ep` of stream `sid` for every lane with `active` set: cut (lists in `col`, finished boxes into `stage`), rank
// by base height (`cnt`: the lane's level counters, see level_count), place into ring row `ep` of bin `bin` with the
// look-ahead copies.  Returns the number of boxes; gave_up: a list or the staging column was full -- nothing written.
template <int FB, class Col>
__device__ __forceinline__ int rows_cut_and_place(const bpp_stream &s, bool active, int bin, int ep, uint64_t sid, const Col col, const Col stage,
                                                  uint32_t *cnt, int cap, int S, bool &gave_up) {
    const int E = s.num_envs, T = s.pool_len, D = s.depth, H = s.H;
    const uint32_t lo = (uint32_t)s.bound_lo, hi = (uint32_t)s.bound_hi;
    const uint32_t whole = (uint32_t)s.W | ((uint32_t)s.L << FB) | ((uint32_t)s.H << (2 * FB));
    const uint32_t term = (uint32_t)s.W | ((uint32_t)s.L << 8) | ((uint32_t)s.H << 16);
    uint32_t *row = (uint32_t *)s.ring + ((size_t)(ep % D) * E + bin) * T + kRowHdr;
    CtrLane c{whole, 0, 1, 0, 0, 0, row, CtrRng::key(s.seed0, sid, (uint32_t)ep)};
    if (active) col.set(0u, whole);
    const int NW = (H + 2) >> 1;                      // words of level counters: levels 0 .. H - 1 and the one for nothing
    for (int l = 0; l < NW; ++l) cnt[l * 64] = 0u;
    const EmitToStage<Col, FB> emit{stage, cnt, (uint32_t)S, (uint32_t)H};
    const bool mine0 = active;
    gave_up = false;
    for (;;) {
        const bool full = active && (c.tail_a + 2 > cap || c.tail_b + 1 > cap || c.nv + 2 > S);
        gave_up = gave_up || full;
        active = active && !full;
        if (!__ballot(active)) break;
        if (cut_visit_ctr<FB>(c, col, cap, active ? ~0u : 0u, lo, hi, emit) != 0u) active = false;
    }
    const bool mine = mine0 && !gave_up;
    const int nv = mine ? c.nv : 0;
    // stable counting sort of the lane's boxes by base height: the counts become first positions (four levels, then four
    // boxes, at a time: the LDS round trips of a group overlap), then every box takes the next position of its level
    uint32_t run = 0;
    for (int l0 = 0; l0 < NW; l0 += 4) {
        uint32_t n[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) n[j] = cnt[min(l0 + j, NW - 1) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t even = n[j] & 0xffffu, odd = n[j] >> 16;
            if (l0 + j < NW) cnt[(l0 + j) * 64] = run | ((run + even) << 16);
            run += even + odd;
        }
    }
    uint32_t first = term, second = term;             // the row's first two items: the look-ahead copies in the two rows before
    for (int k0 = 0; __ballot(k0 < nv); k0 += 4) {
        uint32_t b[4], dest[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = stage.get((uint32_t)min(k0 + j, S));
#pragma unroll
        for (int j = 0; j < 4; ++j) dest[j] = level_count(cnt, k0 + j < nv ? b[j] >> (3 * FB) : (uint32_t)H);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t v = cut_box_bytes(b[j], FB) & 0x00ffffffu;
            if (k0 + j < nv) {
                row[dest[j]] = v;
                first = dest[j] == 0u ? v : first;
                second = dest[j] == 1u ? v : second;
            }
        }
    }
    if (mine) {
        if (ep >= 1) ((uint32_t *)s.ring + ((size_t)((ep - 1) % D) * E + bin) * T)[0] = second;
        if (ep >= 2) ((uint32_t *)s.ring + ((size_t)((ep - 2) % D) * E + bin) * T)[1] = first;
    }
    return nv;
}

template <int FB>
__global__ __launch_bounds__(64) void stream_cut_rows_kernel(bpp_stream s, StreamWork w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using LT = typename ListCol<FB>::T;
    const int lane = threadIdx.x;
    const int E = s.num_envs, T = s.pool_len, D = s.depth, cap = w.cap, S = w.stage;
    const int TI = T - kRowHdr;
    uint32_t *cnt = (uint32_t *)smem + lane;          // level counters first, then the lists and the staging columns
    LT *lists = (LT *)(smem + (size_t)((s.H + 2) / 2) * 64 * 4);
    const ListCol<FB> col{lists + lane};
    const ListCol<FB> stage{lists + (size_t)(2 * cap + 1) * 64 + lane};
    // the same memory for a lane that is served alone: lists and staging column of maxn + 2 entries each, stride 1
    const int capx = w.maxn + 2;
    const ListCol<FB, 1> colx{lists};
    const ListCol<FB, 1> stagex{lists + (size_t)(2 * capx + 1)};
    const uint32_t term = (uint32_t)s.W | ((uint32_t)s.L << 8) | ((uint32_t)s.H << 16);
    const int nrows = w.hdr[3];
    for (int q0 = (int)blockIdx.x * 64; q0 < nrows; q0 += (int)gridDim.x * 64) {     // wave-uniform
        const int q = q0 + lane;
        const bool job = q < nrows;
        const int64_t id = job ? w.rows[q] : 0;
        const int bin = (int)(uint32_t)id, ep = (int)(id >> 32);
        uint64_t sid = 0;
        if (job) {
            const uint32_t *rc = s.mt + (size_t)bin * kCtrRec;
            sid = (uint64_t)rc[0] | ((uint64_t)rc[1] << 32);
            if (ep + 1 == w.target[bin]) s.gen_next[bin] = ep + 1;       // the bin's newest row: its generator is now here
        }
        bool gave_up;
        int nv = rows_cut_and_place<FB>(s, job, bin, ep, sid, col, stage, cnt, cap, S, gave_up);
        uint64_t again = __ballot(gave_up);
        while (again) {                               // rare: one lane at a time with all of the wave's LDS
            const int r = __ffsll((unsigned long long)again) - 1;
            again &= again - 1;
            wave_sync();
            bool never;
            const int n2 = rows_cut_and_place<FB>(s, lane == r, bin, ep, sid, colx, stagex, cnt, capx, capx, never);
            if (lane == r) nv = n2;
            wave_sync();
        }
        // terminator padding: the wave, row by row
        uint32_t *row = (uint32_t *)s.ring + ((size_t)(ep % D) * E + bin) * T + kRowHdr;
        const uint32_t plo = (uint32_t)(uintptr_t)row, phi = (uint32_t)((uint64_t)(uintptr_t)row >> 32);
        uint64_t todo = __ballot(job);
        while (todo) {
            const int r = __ffsll((unsigned long long)todo) - 1;
            todo &= todo - 1;
            const int nr = (int)__builtin_amdgcn_readlane((uint32_t)nv, r);
            const uint32_t rlo = (uint32_t)__builtin_amdgcn_readlane(plo, r), rhi = (uint32_t)__builtin_amdgcn_readlane(phi, r);   // (the builtin returns int)
            uint32_t *rr = (uint32_t *)(uintptr_t)((uint64_t)rlo | ((uint64_t)rhi << 32));
            for (int k = nr + lane; k < TI; k += 64) rr[k] = term;
        }
    }
}

__global__ __launch_bounds__(256) void stream_sort_kernel(bpp_stream s, StreamWork w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = s.num_envs, T = s.pool_len, D = s.depth;
    const int TI = T - kRowHdr;                       // entries of a row behind its look-ahead header
    uint32_t *ent = (uint32_t *)smem + (size_t)wave * (T + 256);
    int *lvl = (int *)(ent + T);                      // per base height: count, then first free position
    const uint32_t term = (uint32_t)s.W | ((uint32_t)s.L << 8) | ((uint32_t)s.H << 16);
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int nrows = w.hdr[3], stride = gridDim.x * 4;
    int q = blockIdx.x * 4 + wave;                    // wave-uniform
    if (q >= nrows) return;
    auto base_of = [&](int bin, int episode) { return (uint32_t *)s.ring + ((size_t)(episode % D) * E + bin) * T; };
    const int fb = w.fb;
    int64_t id = w.rows[q];
    uint32_t *row = base_of((int)(uint32_t)id, (int)(id >> 32)) + kRowHdr;
    uint32_t mine = lane < TI - 1 ? cut_box_bytes(row[lane], fb) : 0u;    // this lane's box if the row is short, and the row's length
    int nv = (int)row[TI - 1];
    for (;;) {
        const int qn = q + stride;
        int64_t idn = id;
        uint32_t *rown = row;
        uint32_t minen = 0;
        int nvn = 0;
        if (qn < nrows) {                             // next row: loads in flight while this one is ranked
            idn = w.rows[qn];
            rown = base_of((int)(uint32_t)idn, (int)(idn >> 32)) + kRowHdr;
            minen = lane < TI - 1 ? cut_box_bytes(rown[lane], fb) : 0u;
            nvn = (int)rown[TI - 1];
        }
        // the look-ahead copies of this row's first two items (terminators where the row is shorter): item 1 goes into entry
        // 0 of the row BEFORE this one, item 0 into entry 1 of the row two before -- written by the lanes that place them
        const int bin = (int)(uint32_t)id, ep = (int)(id >> 32);
        uint32_t *hdr1 = ep >= 1 ? base_of(bin, ep - 1) : nullptr, *hdr0 = ep >= 2 ? base_of(bin, ep - 2) + 1 : nullptr;
        if (lane == 0) {
            if (nv < 2 && hdr1) *hdr1 = term;
            if (nv < 1 && hdr0) *hdr0 = term;
        }
        if (nv <= 64) {
            const bool valid = lane < nv;
            const uint32_t key = mine >> 24;
            int place = 0;
            uint64_t todo = __ballot(valid);
            while (todo) {                            // wave-uniform: one round per distinct base height
                const int l0 = __ffsll((unsigned long long)todo) - 1;
                const uint32_t lv = __builtin_amdgcn_readlane(key, l0);
                const uint64_t m = __ballot(valid && key == lv);
                place += key > lv ? __popcll(m) : (key == lv ? __popcll(m & below) : 0);
                todo &= ~m;
            }
            if (valid) {
                const uint32_t v = mine & 0x00ffffffu;
                row[place] = v;
                if (place == 0 && hdr0) *hdr0 = v;
                if (place == 1 && hdr1) *hdr1 = v;
            }
        } else {
            wave_sync();
            for (int k = lane; k < 256; k += 64) lvl[k] = 0;
            for (int k = lane; k < nv; k += 64) ent[k] = cut_box_bytes(row[k], fb);
            wave_sync();
            for (int k = lane; k < nv; k += 64) atomicAdd(&lvl[ent[k] >> 24], 1);
            wave_sync();
            {   // exclusive prefix over the 256 levels, four per lane
                const int c0 = lvl[4 * lane], c1 = lvl[4 * lane + 1], c2 = lvl[4 * lane + 2], c3 = lvl[4 * lane + 3];
                int incl = c0 + c1 + c2 + c3;
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_up(incl, d, 64);
                    if (lane >= d) incl += o;
                }
                const int ex = incl - (c0 + c1 + c2 + c3);
                wave_sync();
                lvl[4 * lane] = ex;
                lvl[4 * lane + 1] = ex + c0;
                lvl[4 * lane + 2] = ex + c0 + c1;
                lvl[4 * lane + 3] = ex + c0 + c1 + c2;
            }
            wave_sync();
            for (int k0 = 0; k0 < nv; k0 += 64) {     // stable: chunks in order, lanes in order within a level
                const int k = k0 + lane;
                const bool valid = k < nv;
                const uint32_t val = valid ? ent[k] : 0u;
                const uint32_t key = val >> 24;
                int dest = 0;
                uint64_t todo = __ballot(valid);
                while (todo) {
                    const int l0 = __ffsll((unsigned long long)todo) - 1;
                    const uint32_t lv = __builtin_amdgcn_readlane(key, l0);
                    const uint64_t m = __ballot(valid && key == lv);
                    const int first = lvl[lv];
                    if (valid && key == lv) dest = first + __popcll(m & below);
                    wave_sync();
                    if (lane == l0) lvl[lv] = first + __popcll(m);
                    wave_sync();
                    todo &= ~m;
                }
                if (valid) {
                    const uint32_t v = val & 0x00ffffffu;
                    row[dest] = v;
                    if (dest == 0 && hdr0) *hdr0 = v;
                    if (dest == 1 && hdr1) *hdr1 = v;
                }
            }
        }
        for (int k = nv + lane; k < TI; k += 64) row[k] = term;
        if (qn >= nrows) break;
        q = qn;
        id = idn;
        row = rown;
        mine = minen;
        nv = nvn;
    }
}

// ======================================================================================================================
// Refill, plain version: one lane per bin runs cut2_generate as it stands (any geometry, any row length; knob
// stream_legacy).  Lists and a 32-word output buffer in LDS, beyond that the caller's `work` array.
// ======================================================================================================================
constexpr int kStreamLanes = 64;        // threads per workgroup of the refill kernel
constexpr int kStreamPendCap = 48;
constexpr int kStreamValCap = 64;
constexpr int kStreamRngBuf = 32;
constexpr int kStreamLdsWords = 2 * kStreamPendCap + kStreamValCap + kStreamRngBuf;   // per lane

struct LdsWork {
    uint32_t *lds;      // this lane's column of the pending area: word w at lds[w * 64]
    CutBox *spill;      // global: entry i >= cap at spill[(i - cap) * stride]
    size_t stride;
    __device__ CutBox get(int i) const {
        if (i < kStreamPendCap) return CutBox{lds[(2 * i) * kStreamLanes], lds[(2 * i + 1) * kStreamLanes]};
        return spill[(size_t)(i - kStreamPendCap) * stride];
    }
    __device__ void set(int i, CutBox v) {
        if (i < kStreamPendCap) {
            lds[(2 * i) * kStreamLanes] = v.a;
            lds[(2 * i + 1) * kStreamLanes] = v.b;
        } else {
            spill[(size_t)(i - kStreamPendCap) * stride] = v;
        }
    }
};
struct LdsVals {
    uint32_t *lds;      // this lane's column of the value area
    uint32_t *row;      // global: the pool row itself takes what does not fit (entry i at row[i])
    int cap;            // entries of the row that may be written (T - 1)
    __device__ uint32_t get(int i) const { return i < kStreamValCap ? lds[i * kStreamLanes] : (i < cap ? row[i] : 0u); }
    __device__ void set(int i, uint32_t v) {
        if (i < kStreamValCap) lds[i * kStreamLanes] = v;
        else if (i < cap) row[i] = v;
    }
};

// CPython's random.Random for one bin: the current state in the bin's record, tempered outputs handed out from an LDS
// buffer (a refill of the buffer never crosses the end of the state, so unused outputs are returned by stepping the
// index back).
struct BufferedMT {
    uint32_t *rec;      // the bin's record
    uint32_t *mt;       // the current state: rec + kMtRaw + par * 624
    uint32_t *buf;      // this lane's column of the output buffer
    int idx;            // next state word to temper (0..624)
    int have, pos;      // buffered outputs, next one to hand out
    uint32_t par, next_ok;
    __device__ void twist() {
        idx = 0;
        if (next_ok) {  // a fast refill's pretwist kernel has already made the successor
            par ^= 1u;
            mt = rec + kMtRaw + par * kMtHalf;
            next_ok = 0u;
            return;
        }
        for (int k0 = 0; k0 < 624; k0 += 16) {          // in place, sixteen words per round of loads
            uint32_t cur[17], far[16];
#pragma unroll
            for (int q = 0; q < 17; ++q) {
                const int k = k0 + q;
                cur[q] = mt[k < 624 ? k : 0];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = k0 + q;
                far[q] = k < 624 ? mt[k + 397 < 624 ? k + 397 : k - 227] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = k0 + q;
                if (k < 624) {
                    const uint32_t y = (cur[q] & 0x80000000u) | (cur[q + 1] & 0x7fffffffu);
                    const uint32_t nw = far[q] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                    mt[k] = nw;                          // and its output byte, for the fast pipeline's readers
                    const uint8_t tb = (uint8_t)(mt_temper(nw) >> 24);
                    mt_out8(rec)[par * kMtHalf + k] = tb;
                    if (par == 0u && k < kMtMirrorLen) mt_out8(rec)[2 * kMtHalf + k] = tb;
                }
            }
        }
    }
    __device__ uint32_t u32() {
        if (pos >= have) {
            if (idx >= 624) twist();
            const int n = 624 - idx < kStreamRngBuf ? 624 - idx : kStreamRngBuf;
            uint32_t y[kStreamRngBuf];
#pragma unroll
            for (int q = 0; q < kStreamRngBuf; ++q) y[q] = q < n ? mt[idx + q] : 0u;
#pragma unroll
            for (int q = 0; q < kStreamRngBuf; ++q)
                if (q < n) buf[q * kStreamLanes] = mt_temper(y[q]);
            idx += n;
            have = n;
            pos = 0;
        }
        return buf[(pos++) * kStreamLanes];
    }
    __device__ uint32_t below(uint32_t n) { return getrandbits_below(*this, n); }
};

// One lane per bin: seed the bin's generator with random.Random(seed0 + global id); nothing generated yet.
__global__ __launch_bounds__(256) void stream_init_kernel(bpp_stream s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.num_envs) return;
    if (s.rng == BPP_STREAM_RNG_COUNTER) {   // the record is the bin's stream id
        const uint64_t sid = (uint64_t)(s.env_id_base + e);
        uint32_t *rec = s.mt + (size_t)e * kCtrRec;
        rec[0] = (uint32_t)sid, rec[1] = (uint32_t)(sid >> 32), rec[2] = 0u, rec[3] = 0u;
        s.gen_next[e] = 0;
        return;
    }
    uint32_t *rec = s.mt + (size_t)e * kMtRec;
    StridedMT rng{rec + kMtRaw, 1, 624};     // into the first half; a freshly seeded state has no unused output (index 624)
    rng.seed(s.seed0 + (uint64_t)(s.env_id_base + e));
    rec[kMtPos] = (uint32_t)rng.idx;
    rec[kMtPar] = 0u;
    rec[kMtNextOk] = 0u;
    s.gen_next[e] = 0;
}

// The plain kernel's hand-over of one cut sequence (`n` boxes in `vals`, sorted) to ring row `g` of bin `e`: items behind
// the two look-ahead entries, terminator padding, and this row's first two items into the headers of the two rows before.
__device__ __forceinline__ int stream_store_row(const bpp_stream &s, int e, int g, const LdsVals &vals, int n, uint32_t term) {
    const int E = s.num_envs, T = s.pool_len, D = s.depth, TI = T - kRowHdr;
    uint32_t *row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T + kRowHdr;
    const int nw = n < TI - 1 ? n : TI - 1;
    for (int t = 0; t < nw; ++t) row[t] = vals.get(t) & 0x00ffffffu;   // drop the sort key
    for (int t = nw; t < TI; ++t) row[t] = term;                        // pad with the terminator (last entry always)
    const uint32_t i0 = nw > 0 ? vals.get(0) & 0x00ffffffu : term, i1 = nw > 1 ? vals.get(1) & 0x00ffffffu : term;
    if (g >= 1) ((uint32_t *)s.ring + ((size_t)((g - 1) % D) * E + e) * T)[0] = i1;
    if (g >= 2) ((uint32_t *)s.ring + ((size_t)((g - 2) % D) * E + e) * T)[1] = i0;
    return n > TI - 1;
}

// One lane per bin: cut new sequences into the ring until the bin has `depth` episodes available from its current
// one (rows of episodes the bin has finished are the ones overwritten).
__global__ __launch_bounds__(kStreamLanes) void stream_refill_kernel(bpp_stream s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds = (uint32_t *)smem;
    const int lane = threadIdx.x;
    const int e = blockIdx.x * kStreamLanes + lane;
    if (e >= s.num_envs) return;     // no workgroup-level synchronisation below
    const int E = s.num_envs, D = s.depth;
    const int T = s.pool_len;
    const int cur = s.state[e].episode;
    int g = s.gen_next[e];
    if (g >= cur + D) return;
    const uint32_t term = (uint32_t)s.W | ((uint32_t)s.L << 8) | ((uint32_t)s.H << 16);
    LdsWork work{lds + lane, (CutBox *)s.work + e, (size_t)E};
    int over = 0;
    if (s.rng == BPP_STREAM_RNG_COUNTER) {   // every sequence a fresh generator keyed by (seed0, stream id, episode)
        const uint32_t *rc = s.mt + (size_t)e * kCtrRec;
        const uint64_t sid = (uint64_t)rc[0] | ((uint64_t)rc[1] << 32);
        while (g < cur + D) {
            uint32_t *row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T + kRowHdr;
            LdsVals vals{lds + (2 * kStreamPendCap) * kStreamLanes + lane, row, T - 1 - kRowHdr};
            CtrRng rng = CtrRng::key(s.seed0, sid, (uint32_t)g);
            const int n = cut2_generate(rng, work, vals, s.W, s.L, s.H, s.bound_lo, s.bound_hi);
            over += stream_store_row(s, e, g, vals, n, term);
            ++g;
        }
        s.gen_next[e] = g;
        if (over && s.overflow) atomicAdd(s.overflow, over);
        return;
    }
    uint32_t *rec = s.mt + (size_t)e * kMtRec;
    const uint32_t par0 = rec[kMtPar];
    BufferedMT rng{rec, rec + kMtRaw + par0 * kMtHalf, lds + (2 * kStreamPendCap + kStreamValCap) * kStreamLanes + lane,
                   (int)rec[kMtPos], 0, 0, par0, rec[kMtNextOk]};
    while (g < cur + D) {
        uint32_t *row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T + kRowHdr;
        LdsVals vals{lds + (2 * kStreamPendCap) * kStreamLanes + lane, row, T - 1 - kRowHdr};
        const int n = cut2_generate(rng, work, vals, s.W, s.L, s.H, s.bound_lo, s.bound_hi);
        over += stream_store_row(s, e, g, vals, n, term);
        ++g;
    }
    rec[kMtPos] = (uint32_t)(rng.idx - (rng.have - rng.pos));             // buffered but unused outputs are handed out again
    rec[kMtPar] = rng.par;
    rec[kMtNextOk] = rng.next_ok;
    s.gen_next[e] = g;
    if (over && s.overflow) atomicAdd(s.overflow, over);
}
