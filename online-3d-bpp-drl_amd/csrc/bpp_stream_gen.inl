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
__host__ __device__ inline uint32_t rand_below(Rng &rng, uint32_t n) {   // Random._randbelow_with_getrandbits(n), 0 < n < 2^32
    int k = 0;
    for (uint32_t v = n; v; v >>= 1) ++k;
    uint32_t x = rng.u32() >> (32 - k);
    while (x >= n) x = rng.u32() >> (32 - k);
    return x;
}

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
            const int f = flags[rand_below(rng, (uint32_t)nf)];   // random.choice, :68
            int s1[5], s2[5];                   // x, y, z, low, high of the two parts
            if (f == 0) {                       // :70-79
                if (bx <= lo) continue;
                const int r = 1 + (int)rand_below(rng, (uint32_t)bx);   // random.randint(1, x)
                if (r < lo || bx - r < lo) continue;
                s1[0] = r, s1[1] = by, s1[2] = bz, s1[3] = low, s1[4] = high;
                s2[0] = bx - r, s2[1] = by, s2[2] = bz, s2[3] = low, s2[4] = high;
            } else if (f == 1) {                // :80-89
                if (by < lo) continue;
                const int r = 1 + (int)rand_below(rng, (uint32_t)by);
                if (r < lo || by - r < lo) continue;
                s1[0] = bx, s1[1] = r, s1[2] = bz, s1[3] = low, s1[4] = high;
                s2[0] = bx, s2[1] = by - r, s2[2] = bz, s2[3] = low, s2[4] = high;
            } else {                            // :90-99
                if (bz < lo) continue;
                const int r = 1 + (int)rand_below(rng, (uint32_t)bz);
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
// bin, contiguous.  Two halves of 624 state words: the CURRENT MT19937 state and, when kMtNextOk is set, the state that
// follows it (already twisted -- the fast pipeline regenerates states in a separate, fully parallel kernel so that a
// generator running off the end of its state just changes halves).  Then the index of the next unused word of the
// current state (0..624), which half is current, the flag, and up to 32 tempered outputs carried over from the last
// refill (the cut kernel hands out outputs from a 32-word window and keeps what it did not use).
constexpr int kMtHalf = 624;
constexpr int kMtIdx = 2 * kMtHalf, kMtPar = kMtIdx + 1, kMtNextOk = kMtIdx + 2, kMtLeftN = kMtIdx + 3, kMtLeft = kMtIdx + 4;
constexpr int kMtRec = 1312;            // 2 * 624 + 4 + 32 = 1284, padded to whole 128-byte lines

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
//   pretwist one wave per job whose generator has no successor state yet: the next 624 state words, coalesced, through LDS;
//   cut      one lane per job, single-wave workgroups.  The list walk of mdCreator.py:117-135 is run as a state
//            machine that consumes EXACTLY ONE 32-bit generator output per iteration (a rejected randbelow draw, a
//            failed split attempt and a split are all one iteration), so the lanes of a wave stay converged, read
//            their outputs from the same slot of a 32-word LDS window and refill that window together (the raw words of
//            the next window are loaded one window ahead).  The pending boxes live in two LDS lists per lane (this
//            pass / survivors for the next pass: no shifting, no compaction).  The iteration is written without
//            divergent branches -- a wave alone on its SIMD pays ~13 cycles per instruction in branchy code, and every
//            path is taken by some lane in every iteration anyway: stores that do not apply go to a dummy slot.  Cut
//            boxes go straight into the ring row, unsorted, with their base height as sort key;
//   sort     one wave per rewritten row: stable counting sort by base height (depart_box, :137-138), key stripped,
//            terminator padding.
// Same draws in the same order as cut2_generate above (tests/test_stream_supply.py runs both against the oracle's
// generator and Python's random module).
// ======================================================================================================================
struct StreamWork {        // views into bpp_stream.work (see plan_stream)
    int32_t *hdr;          // [16]: jobs in bucket 0..2, rows to sort
    int32_t *jobs;         // [3][E]: local bin ids needing 1 / 2 / >= 3 sequences
    int32_t *target;       // [E]: gen_next every bin is brought to (the scan's reading of episode + depth: the step
                           //      kernels may be running beside the refill, the kernels must agree on one value)
    int64_t *rows;         // [D * E]: bin | episode << 32 of every row rewritten by this refill
    uint32_t *spill;       // [2 * nsp][nslots]: pending boxes beyond the LDS lists (rare)
    int32_t cap, nsp, nslots, maxn;
    int32_t kmax, urgent;  // kmax > 0: a bin gets at most kmax sequences per refill unless that leaves it fewer than
                           // `urgent` rows from its current episode (then as many as it takes); 0: always all depth rows
};
constexpr int kRngWin = 32;            // generator outputs per window
constexpr int kTwistWords = 640;       // LDS scratch of a wave-wide twist (624 used)
constexpr int kSortMaxT = 2048;        // longest row the sort kernel stages in LDS
constexpr int kScanThreads = 1024;

// LDS entries per pending list for sequences of at most maxn boxes.  Measured peaks: 10^3 bins 10 on average, above 16
// in 0.5 % of the sequences, 21 at most in 3000; 20^3 bins 68.  Longer lists continue in global memory (the wave then
// runs its general iteration).  10^3: 24 entries -> 23.3 KB per workgroup, which fits beside seven step-kernel
// workgroups on a CU.
__host__ inline int stream_pend_cap(int maxn) {
    int c = (maxn / 18 + 16 + 7) / 8 * 8;
    c = c < 16 ? 16 : (c > 80 ? 80 : c);
    return c < maxn ? c : maxn;
}
__host__ __device__ inline int stream_cut_lds_words(int cap) { return (2 * cap + 1 + kRngWin) * 64 + kTwistWords; }

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
        const int base = tot ? atomicAdd(&w.hdr[threadIdx.x], tot) : 0;
        for (int k = 0; k < kScanThreads / 64; ++k) wave_base[k][threadIdx.x] += base;
    }
    __syncthreads();
    if (bucket >= 0) w.jobs[(size_t)bucket * E + wave_base[wave][bucket] + __popcll(m[bucket] & below)] = e;
    const int first = g0 + need - nr, at = wave_base[wave][3] + incl - nr;
    for (int k = 0; k < nr; ++k) w.rows[at + k] = (int64_t)(uint32_t)e | ((int64_t)(first + k) << 32);
}

// The whole wave computes the 624 state words that follow `src` into `dst` (both global, may be the same; tw: LDS
// scratch).  Word k needs the OLD words k and k+1 and word k+397 (old for k < 227, else the NEW word k-227); walking k in
// rounds of 64 consecutive words with all reads of a round before its writes gives every lane exactly those values
// (word 623 reads the new word 0, as the serial loop does).
__device__ __forceinline__ void stream_wave_twist(const uint32_t *src, uint32_t *dst, uint32_t *tw, int lane) {
    for (int k = lane; k < 624; k += 64) tw[k] = src[k];
    wave_sync();
    for (int k0 = 0; k0 < 624; k0 += 64) {
        const int k = k0 + lane;
        uint32_t a = 0, b = 0, c = 0;
        if (k < 624) {
            a = tw[k];
            b = tw[k + 1 < 624 ? k + 1 : 0];
            c = tw[k + 397 < 624 ? k + 397 : k - 227];
        }
        wave_sync();
        if (k < 624) {
            const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
            tw[k] = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        wave_sync();
    }
    for (int k = lane; k < 624; k += 64) dst[k] = tw[k];
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
            stream_wave_twist(rec + par * kMtHalf, rec + (par ^ 1u) * kMtHalf, scratch[wave], lane);
            if (lane == 0) rec[kMtNextOk] = 1u;
        }
        wave_sync();
    }
}

// the pending lists of one lane with their continuation in global memory: list r (0 / 1), entry i
struct PendLists {
    uint32_t *lds;      // &lds[lane]: word w of this lane at lds[w * 64]
    uint32_t *spill;    // &spill[slot]: entry k of list r beyond the LDS part at spill[(r * nsp + k) * nslots]
    int cap, nsp;
    size_t nslots;
    __device__ __forceinline__ uint32_t get(int r, int i) const {
        return i < cap ? lds[(r * cap + i) * 64] : spill[(size_t)(r * nsp + i - cap) * nslots];
    }
    __device__ __forceinline__ void set(int r, int i, uint32_t v) const {
        if (i < cap) lds[(r * cap + i) * 64] = v;
        else spill[(size_t)(r * nsp + i - cap) * nslots] = v;
    }
};

// State of one lane's list walk.  A pending box is x | y << 8 | z << 16 | base height << 24 (its top is base + z).
struct CutLane {
    uint32_t box, v;        // box being visited; side being cut (valid in state 1)
    int st, f;              // 0: the next output chooses the side (random.choice), 1: it is the cut position (randint)
    int i, tail_a, tail_b;  // position in this pass's list, its length, length of the survivors' list
    int side;               // which LDS list is this pass's (0 / 1)
    int nv;                 // boxes cut so far
    uint32_t *row;
};

// One output, general form: any list length (entries beyond the LDS part live in global memory).  Returns true when the
// sequence is complete.  mdCreator.py line numbers as in cut2_generate.
__device__ __forceinline__ bool cut_step_general(CutLane &c, const PendLists &pend, uint32_t u, uint32_t lo, uint32_t hi) {
    const uint32_t bx = c.box & 255u, by = (c.box >> 8) & 255u, bz = (c.box >> 16) & 255u;
    const bool fx = bx > hi, fy = by > hi, fz = bz > hi;            // :60-66
    const uint32_t nf = (uint32_t)fx + (uint32_t)fy + (uint32_t)fz;
    int outcome = 0;                                                  // 1: the attempt failed, 2: split
    uint32_t r = 0;
    if (c.st == 0) {                                                  // random.choice(flags), :68
        const uint32_t x = u >> (nf == 1u ? 31 : 30);                // getrandbits(bit_length(nf))
        if (x < nf) {
            c.f = x == 0u ? (fx ? 0 : (fy ? 1 : 2)) : (x == 1u ? ((fx && fy) ? 1 : 2) : 2);
            c.v = c.f == 0 ? bx : (c.f == 1 ? by : bz);
            if (c.f == 0 ? c.v <= lo : c.v < lo) outcome = 1;         // :71, :81, :91
            else c.st = 1;
        }
    } else {                                                          // random.randint(1, v), :73 / :83 / :93
        const uint32_t x = u >> __clz((int)c.v);                      // getrandbits(bit_length(v))
        if (x < c.v) {
            r = x + 1u;
            outcome = (r < lo || c.v - r < lo) ? 1 : 2;               // :74, :84, :94
        }
    }
    if (outcome == 1) {
        pend.set(c.side ^ 1, c.tail_b++, c.box);                      // stays in invalid_box for the next pass
    } else if (outcome == 2) {
        const uint32_t sh = 8u * (uint32_t)c.f, p1 = c.f == 2 ? c.v - r : r, p2 = c.v - p1;
        const uint32_t rest = c.box & ~(255u << sh);
        const uint32_t c1 = rest | (p1 << sh);
        const uint32_t c2 = (rest | (p2 << sh)) + (c.f == 2 ? p1 << 24 : 0u);   // :97-98: the upper part starts at high - r
        // is_valid (:110-115): the untouched sides are within bounds iff the cut side was the only long one
        if (nf == 1u && p1 <= hi) c.row[c.nv++] = c1;
        else pend.set(c.side, c.tail_a++, c1);                        // appended: visited later in this pass
        if (nf == 1u && p2 <= hi) c.row[c.nv++] = c2;
        else pend.set(c.side, c.tail_a++, c2);
    }
    if (!outcome) return false;
    ++c.i;
    if (outcome == 2 && c.i < c.tail_a) {     // the removal slid the next box under the iterator: not visited in this pass
        pend.set(c.side ^ 1, c.tail_b++, pend.get(c.side, c.i));
        ++c.i;
    }
    bool finished = false;
    if (c.i >= c.tail_a) {                    // end of the `for`: next pass over the survivors, or done
        finished = c.tail_b == 0;
        c.side ^= 1;
        c.tail_a = c.tail_b;
        c.tail_b = 0;
        c.i = 0;
    }
    c.st = 0;
    if (!finished) c.box = pend.get(c.side, c.i);
    return finished;
}

// The same step for lanes whose lists are certain to stay inside LDS (tail_a + 2 <= cap, tail_b + 1 <= cap), written
// without divergent branches: everything is computed, selects pick what applies, and a store that does not apply goes
// to the lane's dummy word.  (`f == 0 ? v <= lo : v < lo`, :71 / :81 / :91, cannot hold: the side was chosen because it
// exceeds hi, and bpp_stream requires hi >= 2 lo - 1 >= lo.)
__device__ __forceinline__ bool cut_step_lds(CutLane &c, uint32_t *col, int cap, bool active, uint32_t u, uint32_t lo, uint32_t hi) {
    const int dummy = 2 * cap;
    const int abase = c.side ? cap : 0, bbase = cap - abase;
    const uint32_t bx = c.box & 255u, by = (c.box >> 8) & 255u, bz = (c.box >> 16) & 255u;
    const bool fx = bx > hi, fy = by > hi, fz = bz > hi;
    const uint32_t nf = (uint32_t)fx + (uint32_t)fy + (uint32_t)fz;
    const bool st0 = c.st == 0;
    const uint32_t x = u >> (st0 ? (nf == 1u ? 31u : 30u) : (uint32_t)__clz((int)c.v));
    const bool acc = active & (x < (st0 ? nf : c.v));
    const int fnew = x == 0u ? (fx ? 0 : (fy ? 1 : 2)) : (x == 1u ? ((fx & fy) ? 1 : 2) : 2);
    const uint32_t vnew = fnew == 0 ? bx : (fnew == 1 ? by : bz);
    const bool choose = acc & st0, fin = acc & !st0;
    const uint32_t r = x + 1u;
    const bool good = (r >= lo) & (c.v - r >= lo);
    const bool split = fin & good, failv = fin & !good;
    const uint32_t sh = 8u * (uint32_t)c.f, p1 = c.f == 2 ? c.v - r : r, p2 = c.v - p1;
    const uint32_t rest = c.box & ~(255u << sh);
    const uint32_t c1 = rest | (p1 << sh);
    const uint32_t c2 = (rest | (p2 << sh)) + (c.f == 2 ? p1 << 24 : 0u);
    const bool only = nf == 1u;
    const bool e1 = split & only & (p1 <= hi), e2 = split & only & (p2 <= hi);
    const bool q1 = split & !e1, q2 = split & !e2;
    if (e1) c.row[c.nv] = c1;
    c.nv += e1;
    if (e2) c.row[c.nv] = c2;
    c.nv += e2;
    col[(failv ? bbase + c.tail_b : dummy) * 64] = c.box;
    c.tail_b += failv;
    col[(q1 ? abase + c.tail_a : dummy) * 64] = c1;
    c.tail_a += q1;
    col[(q2 ? abase + c.tail_a : dummy) * 64] = c2;
    c.tail_a += q2;
    c.st = choose ? 1 : (fin ? 0 : c.st);
    c.f = choose ? fnew : c.f;
    c.v = choose ? vnew : c.v;
    c.i += fin;
    // the two entries after the visited box (the first is skipped after a split, :124) and, written below, the head
    // of the survivors
    const uint32_t n1 = col[(abase + min(c.i, cap - 1)) * 64], n2 = col[(abase + min(c.i + 1, cap - 1)) * 64];
    const bool skip = split & (c.i < c.tail_a);
    col[(skip ? bbase + c.tail_b : dummy) * 64] = n1;
    c.tail_b += skip;
    c.i += skip;
    const uint32_t b0 = col[bbase * 64];
    const bool pass_end = fin & (c.i >= c.tail_a);
    const bool finished = pass_end & (c.tail_b == 0);
    c.box = fin ? (pass_end ? b0 : (skip ? n2 : n1)) : c.box;
    c.side = pass_end ? c.side ^ 1 : c.side;
    c.tail_a = pass_end ? c.tail_b : c.tail_a;
    c.tail_b = pass_end ? 0 : c.tail_b;
    c.i = pass_end ? 0 : c.i;
    return finished;
}

__global__ __launch_bounds__(64) void stream_cut_kernel(bpp_stream s, StreamWork w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds = (uint32_t *)smem;
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
    uint32_t *rec = s.mt + (size_t)e * kMtRec;
    uint32_t *col = lds + lane;                                     // word w of this lane at col[w * 64]
    uint32_t *buf = col + (size_t)(2 * cap + 1) * 64;               // output window: slot q at buf[q * 64]
    uint32_t *tw = lds + (size_t)(2 * cap + 1 + kRngWin) * 64;      // twist scratch of the wave
    const PendLists pend{col, w.spill + (size_t)blockIdx.x * 64 + lane, cap, w.nsp, (size_t)w.nslots};

    int idx = 0, g = 0, need = 0, carried = 0;
    uint32_t par = 0, next_ok = 0;
    if (job) {
        idx = (int)rec[kMtIdx];
        par = rec[kMtPar];
        next_ok = rec[kMtNextOk];
        carried = (int)rec[kMtLeftN];
        g = s.gen_next[e];
        need = w.target[e] - g;
        for (int q = 0; q < kRngWin; ++q)
            if (q < carried) buf[q * 64] = rec[kMtLeft + q];
    }
    bool active = job && need > 0;
    const bool ran = active;
    const uint32_t *cur = rec + par * kMtHalf, *nxt = rec + (par ^ 1u) * kMtHalf;

    // raw state words of the coming window, loaded one window ahead: slot q takes word idx + q - from (from > 0 only
    // for the first window, whose first slots hold the carried outputs); past the end of the current state they come
    // from its successor, which the pretwist kernel prepared -- or, for a job on its second lap, the wave makes it now
    uint32_t raw[kRngWin];
    auto fetch = [&](bool want, int from) {
        uint64_t m = __ballot(want && idx + kRngWin - from > kMtHalf && !next_ok);
        while (m) {                                   // wave-uniform, rare: one bin at a time, all lanes help
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const int el = (int)__builtin_amdgcn_readlane(e, l);
            const uint32_t pl = __builtin_amdgcn_readlane(par, l);
            uint32_t *rl = s.mt + (size_t)el * kMtRec;
            wave_sync();
            stream_wave_twist(rl + pl * kMtHalf, rl + (pl ^ 1u) * kMtHalf, tw, lane);
            wave_sync();
            if (lane == l) next_ok = 1u;
        }
#pragma unroll
        for (int q = 0; q < kRngWin; ++q) {           // unconditional loads from addresses that always exist
            const int k = max(idx + q - from, 0);
            raw[q] = *(k < kMtHalf ? cur + k : nxt + (k - kMtHalf));
        }
    };
    uint32_t *const dummy = col + (size_t)(2 * cap) * 64;
    auto install = [&](bool want, int from) {
#pragma unroll
        for (int q = 0; q < kRngWin; ++q) *((want && q >= from) ? buf + q * 64 : dummy) = mt_temper(raw[q]);
        const int adv = want ? kRngWin - from : 0;
        idx += adv;
        if (idx > kMtHalf) {                          // now drawing from the successor
            idx -= kMtHalf;
            const uint32_t *t = cur;
            cur = nxt;
            nxt = t;
            par ^= 1u;
            next_ok = 0u;
        }
    };
    fetch(active, carried);

    const uint32_t whole = (uint32_t)s.W | ((uint32_t)s.L << 8) | ((uint32_t)s.H << 16);
    CutLane c{whole, 0u, 0, 0, 0, 1, 0, 0, 0, (uint32_t *)s.ring + ((size_t)(active ? g % D : 0) * E + e) * T};
    if (active) col[0] = whole;
    int pos = kRngWin, endpos = 0, from = carried;
    uint32_t un = 0;
    while (__ballot(active)) {
        if (pos == kRngWin) {                         // wave-uniform: every 32 outputs
            install(active, from);
            from = 0;
            fetch(active, 0);
            pos = 0;
            un = buf[0];
        }
        const uint32_t u = un;
        un = buf[(pos + 1 < kRngWin ? pos + 1 : pos) * 64];          // the next iteration's output
        bool finished;
        if (__ballot(active && (c.tail_a + 2 > cap || c.tail_b + 1 > cap))) {   // wave-uniform: a list may leave LDS
            finished = active ? cut_step_general(c, pend, u, lo, hi) : false;
        } else {
            finished = cut_step_lds(c, col, cap, active, u, lo, hi);
        }
        if (finished) {
            c.row[T - 1] = (uint32_t)c.nv;            // length for the sort kernel (which restores the terminator)
            ++g;
            if (--need > 0) {
                c.row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T;
                pend.set(c.side, 0, whole);
                c.tail_a = 1;
                c.nv = 0;
                c.box = whole;
            } else {
                active = false;
                endpos = pos + 1;
            }
        }
        ++pos;
    }
    if (ran) {                                        // keep the outputs of the window that were not used
        rec[kMtIdx] = (uint32_t)idx;
        rec[kMtPar] = par;
        rec[kMtNextOk] = next_ok;
        rec[kMtLeftN] = (uint32_t)(kRngWin - endpos);
        for (int q = 0; q < kRngWin; ++q)
            if (q >= endpos) rec[kMtLeft + q - endpos] = buf[q * 64];
        s.gen_next[e] = g;
    }
}

__global__ __launch_bounds__(256) void stream_sort_kernel(bpp_stream s, StreamWork w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int E = s.num_envs, T = s.pool_len, D = s.depth;
    uint32_t *ent = (uint32_t *)smem + (size_t)wave * (T + 256);
    int *lvl = (int *)(ent + T);                      // per base height: count, then first free position
    const uint32_t term = (uint32_t)s.W | ((uint32_t)s.L << 8) | ((uint32_t)s.H << 16);
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int nrows = w.hdr[3];
    for (int q = blockIdx.x * 4 + wave; q < nrows; q += gridDim.x * 4) {     // wave-uniform
        const int64_t id = w.rows[q];
        const int e = (int)(uint32_t)id, g = (int)(id >> 32);
        uint32_t *row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T;
        const int nv = (int)row[T - 1];
        wave_sync();
        for (int k = lane; k < 256; k += 64) lvl[k] = 0;
        for (int k = lane; k < nv; k += 64) ent[k] = row[k];
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
        for (int k0 = 0; k0 < nv; k0 += 64) {         // stable: chunks in order, lanes in order within a level
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
            if (valid) row[dest] = val & 0x00ffffffu;
        }
        for (int k = nv + lane; k < T; k += 64) row[k] = term;
        wave_sync();
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

// CPython's random.Random for one bin: the bin's record in global memory, tempered outputs handed out from an LDS
// buffer (first the outputs a fast refill left over, then fresh ones; a refill of the buffer never crosses a twist, so
// unused outputs are returned by stepping the index back).
struct BufferedMT {
    uint32_t *mt;       // the current state (one half of the bin's record)
    uint32_t *other;    // the other half: the successor state when next_ok
    uint32_t *buf;      // this lane's column of the output buffer
    int idx;            // next state word to temper (0..624)
    int have, pos;      // buffered outputs, next one to hand out
    bool carried;       // the buffer holds carried-over outputs (not re-derivable from idx)
    uint32_t par, next_ok;
    __device__ void twist() {
        if (next_ok) {  // a fast refill's pretwist kernel has already made the successor
            uint32_t *t = mt;
            mt = other;
            other = t;
            par ^= 1u;
            next_ok = 0u;
            idx = 0;
            return;
        }
        for (int k0 = 0; k0 < 624; k0 += 16) {
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
                    mt[k] = far[q] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
            }
        }
        idx = 0;
    }
    __device__ uint32_t u32() {
        if (pos >= have) {
            carried = false;
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
};

// One lane per bin: seed the bin's generator with random.Random(seed0 + global id); nothing generated yet.
__global__ __launch_bounds__(256) void stream_init_kernel(bpp_stream s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.num_envs) return;
    uint32_t *rec = s.mt + (size_t)e * kMtRec;
    StridedMT rng{rec, 1, 624};          // into the first half
    rng.seed(s.seed0 + (uint64_t)(s.env_id_base + e));
    rec[kMtIdx] = (uint32_t)rng.idx;
    rec[kMtPar] = 0u;
    rec[kMtNextOk] = 0u;
    rec[kMtLeftN] = 0u;
    s.gen_next[e] = 0;
}

// One lane per bin: cut new sequences into the ring until the bin has `depth` episodes available from its current
// one (rows of episodes the bin has finished are the ones overwritten).
__global__ __launch_bounds__(kStreamLanes) void stream_refill_kernel(bpp_stream s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *lds = (uint32_t *)smem;
    const int lane = threadIdx.x;
    const int e = blockIdx.x * kStreamLanes + lane;
    if (e >= s.num_envs) return;     // no workgroup-level synchronisation below
    const int E = s.num_envs, T = s.pool_len, D = s.depth;
    const int cur = s.state[e].episode;
    int g = s.gen_next[e];
    if (g >= cur + D) return;
    uint32_t *rec = s.mt + (size_t)e * kMtRec;
    const uint32_t par0 = rec[kMtPar];
    BufferedMT rng{rec + par0 * kMtHalf, rec + (par0 ^ 1u) * kMtHalf, lds + (2 * kStreamPendCap + kStreamValCap) * kStreamLanes + lane,
                   (int)rec[kMtIdx], (int)rec[kMtLeftN], 0, true, par0, rec[kMtNextOk]};
    for (int q = 0; q < rng.have; ++q) rng.buf[q * kStreamLanes] = rec[kMtLeft + q];
    LdsWork work{lds + lane, (CutBox *)s.work + e, (size_t)E};
    const uint32_t term = (uint32_t)s.W | ((uint32_t)s.L << 8) | ((uint32_t)s.H << 16);
    int over = 0;
    while (g < cur + D) {
        uint32_t *row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T;
        LdsVals vals{lds + (2 * kStreamPendCap) * kStreamLanes + lane, row, T - 1};
        const int n = cut2_generate(rng, work, vals, s.W, s.L, s.H, s.bound_lo, s.bound_hi);
        over += n > T - 1;
        const int nw = n < T - 1 ? n : T - 1;
        for (int t = 0; t < nw; ++t) row[t] = vals.get(t) & 0x00ffffffu;   // drop the sort key
        for (int t = nw; t < T; ++t) row[t] = term;                        // pad with the terminator (last entry always)
        ++g;
    }
    // unused outputs: carried-over ones stay in the record, fresh ones are returned by stepping the index back
    const int left = rng.have - rng.pos;
    if (rng.carried) {
        rec[kMtLeftN] = (uint32_t)left;
        for (int q = 0; q < left; ++q) rec[kMtLeft + q] = rng.buf[(rng.pos + q) * kStreamLanes];
    } else {
        rec[kMtLeftN] = 0u;
        rec[kMtIdx] = (uint32_t)(rng.idx - left);
    }
    rec[kMtPar] = rng.par;
    rec[kMtNextOk] = rng.next_ok;
    s.gen_next[e] = g;
    if (over && s.overflow) atomicAdd(s.overflow, over);
}
