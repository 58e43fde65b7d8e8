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

// pending box: a = x | y << 8 | z << 16, b = low | high << 8   (all <= 255)
struct CutBox {
    uint32_t a, b;
};

template <class Rng>
__host__ __device__ inline uint32_t rand_below(Rng &rng, uint32_t n) {   // Random._randbelow_with_getrandbits(n), 0 < n < 2^32
    int k = 0;
    for (uint32_t v = n; v; v >>= 1) ++k;
    uint32_t x = rng.u32() >> (32 - k);
    while (x >= n) x = rng.u32() >> (32 - k);
    return x;
}

// One CUT-2 sequence into `row` (packed x | y<<8 | z<<16, top byte 0; at most cap items are stored), every box side in
// [lo, hi].  `work(i)` addresses the pending list (capacity >= W*L*H / lo^3 + 8).  Returns the number of items.
template <class Rng, class Work>
__host__ __device__ inline int cut2_generate(Rng &rng, Work &work, int W, int L, int H, int lo, int hi, uint32_t *row, int cap) {
    int nv = 0, ni = 0;
    work(ni++) = CutBox{(uint32_t)W | ((uint32_t)L << 8) | ((uint32_t)H << 16), 0u | ((uint32_t)H << 8)};
    while (ni) {
        int i = 0;
        while (i < ni) {                        // `for box in invalid_box` with remove/append inside, mdCreator.py:121-130
            const CutBox b = work(i++);
            const int bx = b.a & 255u, by = (b.a >> 8) & 255u, bz = (b.a >> 16) & 255u, low = b.b & 255u, high = (b.b >> 8) & 255u;
            int flags[3], nf = 0;               // :60-66
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
            for (int k = i; k < ni; ++k) work(k - 1) = work(k);   // invalid_box.remove(box)
            --ni;
            for (int part = 0; part < 2; ++part) {
                const int *c = part ? s2 : s1;
                const bool ok = c[0] >= lo && c[0] <= hi && c[1] >= lo && c[1] <= hi && c[2] >= lo && c[2] <= hi;
                if (ok) {
                    if (nv < cap) row[nv] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24);
                    ++nv;
                } else {
                    work(ni++) = CutBox{(uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16),
                                        (uint32_t)c[3] | ((uint32_t)c[4] << 8)};
                }
            }
        }
    }
    // depart_box (:137-138): stable sort by the height of the base (kept in the top byte so far), then drop the key
    const int n = nv < cap ? nv : cap;
    for (int a = 1; a < n; ++a) {
        const uint32_t v = row[a];
        int k = a - 1;
        while (k >= 0 && (row[k] >> 24) > (v >> 24)) {
            row[k + 1] = row[k];
            --k;
        }
        row[k + 1] = v;
    }
    for (int a = 0; a < n; ++a) row[a] &= 0x00ffffffu;
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

struct StridedWork {
    CutBox *base;
    size_t stride;
    __host__ __device__ CutBox &operator()(int i) { return base[(size_t)i * stride]; }
};

constexpr int kStreamMtWords = 625;   // 624 state words + the index

__host__ __device__ inline int stream_work_entries(int W, int L, int H, int lo) { return W * L * H / (lo * lo * lo) + 8; }

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
__global__ __launch_bounds__(256) void stream_refill_kernel(bpp_stream s) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.num_envs) return;
    const int E = s.num_envs, T = s.pool_len, D = s.depth;
    const int cur = s.state[e].episode;
    int g = s.gen_next[e];
    if (g >= cur + D) return;
    StridedMT rng{s.mt + e, (size_t)E, (int)s.mt[(size_t)624 * E + e]};
    StridedWork work{(CutBox *)s.work + e, (size_t)E};
    const uint32_t term = (uint32_t)s.W | ((uint32_t)s.L << 8) | ((uint32_t)s.H << 16);
    int over = 0;
    while (g < cur + D) {
        uint32_t *row = (uint32_t *)s.ring + ((size_t)(g % D) * E + e) * T;
        const int n = cut2_generate(rng, work, s.W, s.L, s.H, s.bound_lo, s.bound_hi, row, T - 1);
        over += n > T - 1;
        for (int t = n < T - 1 ? n : T - 1; t < T; ++t) row[t] = term;   // pad with the terminator (last entry always)
        ++g;
    }
    s.mt[(size_t)624 * E + e] = (uint32_t)rng.idx;
    s.gen_next[e] = g;
    if (over && s.overflow) atomicAdd(s.overflow, over);
}
