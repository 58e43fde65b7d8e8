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

// ---- CPython's random.Random on MT19937 with the state words at a stride (device: one bin per lane, word i of
// bin e at mt[i * stride + e], so a wave's accesses are contiguous; host: stride 1) -------------------------------
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

constexpr int kStreamMtWords = 625;   // 624 state words + the index

__host__ __device__ inline int stream_work_entries(int W, int L, int H, int lo) { return W * L * H / (lo * lo * lo) + 8; }

// ---- device side: one lane per bin, the hot state in LDS -------------------------------------------------------------
// Every access of the list walk is a dependent round trip, so where the lists live decides the kernel's speed: the
// first kStreamPendCap pending boxes and kStreamValCap cut boxes of a lane sit in LDS (word w of lane l at
// lds[w * 64 + l]: conflict-free), anything beyond -- rare -- spills to the caller's global `work` array.  The
// generator draws from a 32-word LDS buffer of tempered outputs refilled with 32 independent loads, and twists its
// state in chunks of 16 words (17 + 16 independent loads, then 16 stores) instead of word by word.
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

// CPython's random.Random for one bin: state words in global memory (word i of bin e at mt[i * E + e]), tempered outputs
// handed out from an LDS buffer.
struct BufferedMT {
    uint32_t *mt;       // &mt[e]
    size_t stride;      // E
    uint32_t *buf;      // this lane's column of the output buffer
    int idx;            // next state word to temper (0..624)
    int have, pos;      // buffered outputs, next one to hand out
    __device__ void twist() {
        for (int k0 = 0; k0 < 624; k0 += 16) {
            uint32_t cur[17], far[16];
#pragma unroll
            for (int q = 0; q < 17; ++q) {
                const int k = k0 + q;
                cur[q] = mt[(size_t)(k < 624 ? k : 0) * stride];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = k0 + q;
                far[q] = k < 624 ? mt[(size_t)(k + 397 < 624 ? k + 397 : k - 227) * stride] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int k = k0 + q;
                if (k < 624) {
                    const uint32_t y = (cur[q] & 0x80000000u) | (cur[q + 1] & 0x7fffffffu);
                    mt[(size_t)k * stride] = far[q] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                }
            }
        }
        idx = 0;
    }
    __device__ uint32_t u32() {
        if (pos >= have) {
            if (idx >= 624) twist();
            const int n = 624 - idx < kStreamRngBuf ? 624 - idx : kStreamRngBuf;
            uint32_t y[kStreamRngBuf];
#pragma unroll
            for (int q = 0; q < kStreamRngBuf; ++q) y[q] = q < n ? mt[(size_t)(idx + q) * stride] : 0u;
#pragma unroll
            for (int q = 0; q < kStreamRngBuf; ++q) {
                uint32_t v = y[q];
                v ^= v >> 11;
                v ^= (v << 7) & 0x9d2c5680u;
                v ^= (v << 15) & 0xefc60000u;
                v ^= v >> 18;
                if (q < n) buf[q * kStreamLanes] = v;
            }
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
    StridedMT rng{s.mt + e, (size_t)s.num_envs, 624};
    rng.seed(s.seed0 + (uint64_t)(s.env_id_base + e));
    s.mt[(size_t)624 * s.num_envs + e] = (uint32_t)rng.idx;
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
    BufferedMT rng{s.mt + e, (size_t)E, lds + (2 * kStreamPendCap + kStreamValCap) * kStreamLanes + lane,
                   (int)s.mt[(size_t)624 * E + e], 0, 0};
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
    // buffered but unused outputs are handed out again next time: remember the index of the next unused state word
    s.mt[(size_t)624 * E + e] = (uint32_t)(rng.idx - (rng.have - rng.pos));
    s.gen_next[e] = g;
    if (over && s.overflow) atomicAdd(s.overflow, over);
}
