/* bpp_gen.inl -- shared host implementation of bpp_gen_cut2 (include/bpp_abi.h); included by both
 * csrc/bpp_kernels.hip and oracle/bpp_oracle.c so the two libraries export the same generator.
 * Plain C99-compatible code (also valid C++). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- CPython's random.Random: MT19937 (Modules/_randommodule.c) ---------------------------------- */
typedef struct { uint32_t mt[624]; int idx; } bpp_mt;

static void bpp_mt_init_genrand(bpp_mt *r, uint32_t s) {
    r->mt[0] = s;
    for (int i = 1; i < 624; ++i) r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->idx = 624;
}

/* random.seed(int): init_by_array over the 32-bit little-endian digits of abs(seed) */
static void bpp_mt_seed(bpp_mt *r, uint64_t seed) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    int klen = key[1] ? 2 : 1;
    bpp_mt_init_genrand(r, 19650218u);
    int i = 1, j = 0;
    for (int k = 624 > klen ? 624 : klen; k; --k) {
        r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        if (++i >= 624) { r->mt[0] = r->mt[623]; i = 1; }
        if (++j >= klen) j = 0;
    }
    for (int k = 623; k; --k) {
        r->mt[i] = (r->mt[i] ^ ((r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        if (++i >= 624) { r->mt[0] = r->mt[623]; i = 1; }
    }
    r->mt[0] = 0x80000000u;
    r->idx = 624;
}

static uint32_t bpp_mt_u32(bpp_mt *r) {
    if (r->idx >= 624) {
        uint32_t *mt = r->mt;
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        r->idx = 0;
    }
    uint32_t y = r->mt[r->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* Random._randbelow_with_getrandbits(n), 0 < n < 2^32 */
static uint32_t bpp_mt_below(bpp_mt *r, uint32_t n) {
    int k = 0;
    for (uint32_t v = n; v; v >>= 1) ++k;
    uint32_t x = bpp_mt_u32(r) >> (32 - k);
    while (x >= n) x = bpp_mt_u32(r) >> (32 - k);
    return x;
}

/* ---- counter-based generator of BPP_STREAM_RNG_COUNTER (normative definition: include/bpp_abi.h) ---- */
typedef struct { uint32_t klo, khi, n; } bpp_ctr;

static uint32_t bpp_fmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
static bpp_ctr bpp_ctr_key(uint64_t seed0, uint64_t sid, uint32_t k) {
    uint32_t h = bpp_fmix32((uint32_t)seed0 + 0x9E3779B9u);
    h = bpp_fmix32(h ^ (uint32_t)(seed0 >> 32));
    h = bpp_fmix32(h ^ (uint32_t)sid);
    h = bpp_fmix32(h ^ (uint32_t)(sid >> 32));
    bpp_ctr c;
    c.klo = bpp_fmix32(h ^ k);
    c.khi = bpp_fmix32(c.klo + 0x7F4A7C15u + k);
    c.n = 0;
    return c;
}
static uint32_t bpp_ctr_word(const bpp_ctr *c, uint32_t a) {
    return bpp_fmix32((c->klo + c->n * 0x9E3779B9u) ^ (c->khi + a * 0x85EBCA77u));
}
static uint32_t bpp_ctr_below(bpp_ctr *c, uint32_t lim) {   /* Lemire's unbiased multiply-shift; one draw index per call */
    uint32_t a = 0;
    uint64_t m = (uint64_t)bpp_ctr_word(c, a) * lim;
    if ((uint32_t)m < lim) {
        const uint32_t t = (0u - lim) % lim;
        while ((uint32_t)m < t) m = (uint64_t)bpp_ctr_word(c, ++a) * lim;
    }
    c->n += 1;
    return (uint32_t)(m >> 32);
}

/* the two generators behind one call: `below(rng, n)` */
typedef uint32_t (*bpp_below_fn)(void *, uint32_t);
static uint32_t bpp_below_mt(void *r, uint32_t n) { return bpp_mt_below((bpp_mt *)r, n); }
static uint32_t bpp_below_ctr(void *r, uint32_t n) { return bpp_ctr_below((bpp_ctr *)r, n); }

/* ---- envs/bpp0/mdCreator.py:59-166 ---------------------------------------------------------------- */
typedef struct { int x, y, z, low, high; } bpp_cut;

static int bpp_cmp_low(const void *a, const void *b) {  /* stable sort by low_bound: tie -> original order */
    const bpp_cut *p = (const bpp_cut *)a, *q = (const bpp_cut *)b;
    if (p->low != q->low) return p->low < q->low ? -1 : 1;
    return p->high < q->high ? -1 : (p->high > q->high ? 1 : 0);  /* `high` temporarily holds the original index */
}

/* one sequence drawn through `below`; returns the number of items, writes at most cap of them */
static int bpp_cut2_walk(bpp_below_fn below, void *rngp, int W, int L, int H, int lo, int hi, uint8_t *out, int cap) {
    int vol = W * L * H, maxn = vol / (lo * lo * lo) + 8;
    bpp_cut *valid = (bpp_cut *)malloc(sizeof(bpp_cut) * (size_t)maxn);
    bpp_cut *inv = (bpp_cut *)malloc(sizeof(bpp_cut) * (size_t)maxn * 2);
    int nv = 0, ni = 0;
    inv[ni++] = (bpp_cut){W, L, H, 0, H};
    while (ni) {
        int i = 0;
        while (i < ni) {                        /* `for box in invalid_box` with remove/append inside, :121-130 */
            bpp_cut b = inv[i++];
            int flags[3], nf = 0;               /* :60-66 */
            if (b.x > hi) flags[nf++] = 0;
            if (b.y > hi) flags[nf++] = 1;
            if (b.z > hi) flags[nf++] = 2;
            int f = flags[below(rngp, (uint32_t)nf)];   /* random.choice, :68 */
            bpp_cut s1, s2;
            if (f == 0) {                       /* :70-79 */
                if (b.x <= lo) continue;
                int r = 1 + (int)below(rngp, (uint32_t)b.x);   /* random.randint(1, x) */
                if (r < lo || b.x - r < lo) continue;
                s1 = (bpp_cut){r, b.y, b.z, b.low, b.high};
                s2 = (bpp_cut){b.x - r, b.y, b.z, b.low, b.high};
            } else if (f == 1) {                /* :80-89 */
                if (b.y < lo) continue;
                int r = 1 + (int)below(rngp, (uint32_t)b.y);
                if (r < lo || b.y - r < lo) continue;
                s1 = (bpp_cut){b.x, r, b.z, b.low, b.high};
                s2 = (bpp_cut){b.x, b.y - r, b.z, b.low, b.high};
            } else {                            /* :90-99 */
                if (b.z < lo) continue;
                int r = 1 + (int)below(rngp, (uint32_t)b.z);
                if (r < lo || b.z - r < lo) continue;
                s1 = (bpp_cut){b.x, b.y, b.z - r, b.low, b.high - r};
                s2 = (bpp_cut){b.x, b.y, r, b.high - r, b.high};
            }
            memmove(inv + i - 1, inv + i, sizeof(bpp_cut) * (size_t)(ni - i));   /* invalid_box.remove(box) */
            --ni;
            bpp_cut subs[2] = {s1, s2};
            for (int k = 0; k < 2; ++k) {
                bpp_cut c = subs[k];
                int ok = c.x >= lo && c.x <= hi && c.y >= lo && c.y <= hi && c.z >= lo && c.z <= hi;
                if (ok) valid[nv++] = c; else inv[ni++] = c;
            }
        }
    }
    for (int k = 0; k < nv; ++k) valid[k].high = k;   /* depart_box: stable sort by low_bound, :137-138 */
    qsort(valid, (size_t)nv, sizeof(bpp_cut), bpp_cmp_low);
    for (int k = 0; k < nv && k < cap; ++k) {
        out[4 * k] = (uint8_t)valid[k].x;
        out[4 * k + 1] = (uint8_t)valid[k].y;
        out[4 * k + 2] = (uint8_t)valid[k].z;
        out[4 * k + 3] = 0;
    }
    free(valid);
    free(inv);
    return nv;
}

/* one sequence from a running random.Random stream */
static int bpp_cut2_from_stream(bpp_mt *rngp, int W, int L, int H, int lo, int hi, uint8_t *out, int cap) {
    return bpp_cut2_walk(bpp_below_mt, rngp, W, L, H, lo, hi, out, cap);
}
/* episode k of counter stream (seed0, sid) */
static int bpp_cut2_counter(uint64_t seed0, uint64_t sid, uint32_t k, int W, int L, int H, int lo, int hi, uint8_t *out, int cap) {
    bpp_ctr c = bpp_ctr_key(seed0, sid, k);
    return bpp_cut2_walk(bpp_below_ctr, &c, W, L, H, lo, hi, out, cap);
}

/* sequence of a fresh random.Random(seed) */
static int bpp_cut2_sequence(int W, int L, int H, int lo, int hi, uint64_t seed, uint8_t *out, int cap) {
    bpp_mt r;
    bpp_mt_seed(&r, seed);
    return bpp_cut2_from_stream(&r, W, L, H, lo, hi, out, cap);
}

/* Arguments the reference itself cannot handle: a bin already inside the bounds, or with a side below
 * the lower bound, makes Box.benchmark_split call random.choice([]) (IndexError, mdCreator.py:60-68);
 * bound_hi < 2*bound_lo - 1 leaves pieces that can never be cut into range (endless loop, :117-135). */
static int bpp_gen_cut2_args_ok(int n, int T, int W, int L, int H, int lo, int hi) {
    if (n <= 0 || T < 2 || W <= 0 || L <= 0 || H <= 0 || W > 255 || L > 255 || H > 255) return 0;
    if (lo < 1 || hi < 2 * lo - 1) return 0;
    if (W < lo || L < lo || H < lo) return 0;
    if (W <= hi && L <= hi && H <= hi) return 0;
    return 1;
}

static int bpp_gen_cut2_range(uint8_t *pool, int32_t *lengths, int k0, int k1, int T, int W, int L, int H, int lo, int hi,
                              uint64_t seed0) {
    int overflow = 0;
    for (int k = k0; k < k1; ++k) {
        uint8_t *row = pool + (size_t)k * T * 4;
        for (int t = 0; t < T; ++t) {
            row[4 * t] = (uint8_t)W;
            row[4 * t + 1] = (uint8_t)L;
            row[4 * t + 2] = (uint8_t)H;
            row[4 * t + 3] = 0;
        }
        int n = bpp_cut2_sequence(W, L, H, lo, hi, seed0 + (uint64_t)k, row, T - 1);
        if (lengths) lengths[k] = n;
        if (n > T - 1) overflow = 1;
    }
    return overflow;
}

/* ---- numpy legacy RandomState on the same MT19937 (numpy/random/_mt19937.pyx, legacy seeding) ------ */
/* np.random.seed(int s), 0 <= s < 2^32: mt19937_seed == init_genrand(s) */
static void bpp_npmt_seed(bpp_mt *r, uint32_t s) { bpp_mt_init_genrand(r, s); }
/* np.random.rand(): mt19937_next_double */
static double bpp_npmt_double(bpp_mt *r) {
    uint32_t a = bpp_mt_u32(r) >> 5, b = bpp_mt_u32(r) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}
/* np.random.randint(0, n) (legacy, dtype int64, n - 1 < 2^32): masked rejection on 32-bit draws
 * (numpy/random/src/distributions/distributions.c: buffered_bounded_masked_uint32) */
static uint32_t bpp_npmt_below(bpp_mt *r, uint32_t n) {
    uint32_t rng = n - 1u, mask = rng, v;
    if (rng == 0) return 0;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    do v = bpp_mt_u32(r) & mask; while (v > rng);
    return v;
}

/* ---- envs/bpp0/cutCreator.py:32-128 (CUT-1) --------------------------------------------------------- */
typedef struct { int x, y, z, lx, ly, lz; } bpp_meta;

/* returns the number of items (writes at most cap), or -1 when the reference itself would fail
 * (`assert pos_range[0] <= pos_range[1]`, cutCreator.py:74: a piece smaller than the lower bound) */
static int bpp_cut1_sequence(int W, int L, int H, const int32_t rg[6], int rotation, uint64_t seed, uint8_t *out, int cap) {
    const int lowv[3] = {rg[0], rg[1], rg[2]}, highv[3] = {rg[3], rg[4], rg[5]};
    int vol = W * L * H, minv = (lowv[0] < 1 ? 1 : lowv[0]) * (lowv[1] < 1 ? 1 : lowv[1]) * (lowv[2] < 1 ? 1 : lowv[2]);
    int maxn = vol / minv + 8;
    bpp_meta *cur = (bpp_meta *)malloc(sizeof(bpp_meta) * (size_t)maxn * 2);
    bpp_meta *nxt = (bpp_meta *)malloc(sizeof(bpp_meta) * (size_t)maxn * 2);
    bpp_meta *cand = (bpp_meta *)malloc(sizeof(bpp_meta) * (size_t)maxn * 2);
    int *plain = (int *)calloc((size_t)W * L, sizeof(int));
    int nc = 0, nn = 0, ncand = 0, nout = 0, bad = 0;
    bpp_mt rng, nprng;
    bpp_mt_seed(&rng, seed);
    bpp_npmt_seed(&nprng, (uint32_t)seed);
    cur[nc++] = (bpp_meta){W, L, H, 0, 0, 0};
    int again = 1;
    while (again && !bad) {                                    /* _cut_box, :78-95 */
        again = 0;
        nn = 0;
        for (int i = 0; i < nc && !bad; ++i) {
            bpp_meta b = cur[i];
            int dim[3] = {b.x, b.y, b.z}, df_list[3], nd = 0;
            for (int d = 0; d < 3; ++d)
                if (dim[d] < lowv[d] || dim[d] > highv[d]) df_list[nd++] = d;   /* _check_box, :52-56 */
            if (nd == 0) { nxt[nn++] = b; continue; }
            int df = df_list[bpp_mt_below(&rng, (uint32_t)nd)];                 /* random.choice, :66 */
            int lo = lowv[df], hi = dim[df] - lowv[df];
            if (lo > hi) { bad = 1; break; }                                    /* assert, :74 */
            int pos = lo + (int)bpp_mt_below(&rng, (uint32_t)(hi - lo + 1));    /* random.randint(lo, hi), :75 */
            bpp_meta b1 = b, b2 = b;                                            /* MetaBox.split, :16-26 */
            if (df == 0) { b1.x = pos; b2.x = b.x - pos; b2.lx = b.lx + pos; }
            else if (df == 1) { b1.y = pos; b2.y = b.y - pos; b2.ly = b.ly + pos; }
            else { b1.z = pos; b2.z = b.z - pos; b2.lz = b.lz + pos; }
            if (nn + 2 > maxn * 2) { bad = 1; break; }
            nxt[nn++] = b1;
            nxt[nn++] = b2;
            again = 1;
        }
        bpp_meta *t = cur; cur = nxt; nxt = t;
        nc = nn;
    }
    while (!bad) {
        /* _add_candidate, :97-106: pieces whose whole footprint is at their base height become drawable */
        nn = 0;
        for (int i = 0; i < nc; ++i) {
            bpp_meta m = cur[i];
            int ok = 1;
            for (int a = m.lx; a < m.lx + m.x && ok; ++a)
                for (int c = m.ly; c < m.ly + m.y; ++c)
                    if (plain[a * L + c] != m.lz) { ok = 0; break; }
            if (ok) cand[ncand++] = m; else cur[nn++] = m;
        }
        nc = nn;
        if (ncand == 0) break;                                                  /* generate_box_size, :111-128 */
        int idx = (int)bpp_mt_below(&rng, (uint32_t)ncand);                     /* random.randint(0, len - 1) */
        bpp_meta b = cand[idx];
        memmove(cand + idx, cand + idx + 1, sizeof(bpp_meta) * (size_t)(ncand - idx - 1));   /* candidates.pop(idx) */
        --ncand;
        int ox = b.x, oy = b.y;
        if (rotation && !(bpp_npmt_double(&nprng) < 0.5)) { ox = b.y; oy = b.x; }            /* :119-125 */
        if (nout < cap) {
            out[4 * nout] = (uint8_t)ox;
            out[4 * nout + 1] = (uint8_t)oy;
            out[4 * nout + 2] = (uint8_t)b.z;
            out[4 * nout + 3] = 0;
        }
        ++nout;
        for (int a = b.lx; a < b.lx + b.x; ++a)                                  /* _update, :108-109 */
            for (int c = b.ly; c < b.ly + b.y; ++c) plain[a * L + c] += b.z;
    }
    free(cur);
    free(nxt);
    free(cand);
    free(plain);
    return bad ? -1 : nout;
}

static int bpp_gen_cut1_args_ok(int n, int T, int W, int L, int H, const int32_t rg[6]) {
    if (n <= 0 || T < 2 || W <= 0 || L <= 0 || H <= 0 || W > 255 || L > 255 || H > 255 || !rg) return 0;
    for (int d = 0; d < 3; ++d)
        if (rg[d] < 1 || rg[d + 3] < 2 * rg[d] - 1) return 0;   /* a piece in (high, 2*low) could never be cut into range */
    if (W < rg[0] || L < rg[1] || H < rg[2]) return 0;
    return 1;
}

/* rows k0..k1-1; returns 0, 1 (a sequence does not fit) or 2 (the reference's assert would fire) */
static int bpp_gen_cut1_range(uint8_t *pool, int32_t *lengths, int k0, int k1, int T, int W, int L, int H, const int32_t rg[6],
                              int rotation, uint64_t seed0) {
    int status = 0;
    for (int k = k0; k < k1; ++k) {
        uint8_t *row = pool + (size_t)k * T * 4;
        for (int t = 0; t < T; ++t) {
            row[4 * t] = (uint8_t)W;
            row[4 * t + 1] = (uint8_t)L;
            row[4 * t + 2] = (uint8_t)H;
            row[4 * t + 3] = 0;
        }
        int n = bpp_cut1_sequence(W, L, H, rg, rotation, seed0 + (uint64_t)k, row, T - 1);
        if (lengths) lengths[k] = n;
        if (n < 0) status = 2;
        else if (n > T - 1 && status == 0) status = 1;
    }
    return status;
}

/* ---- envs/bpp0/binCreator.py:24-40 (RS): T-1 items uniform over box_set, then the terminator -------- */
static void bpp_gen_rs_range(uint8_t *pool, int k0, int k1, int T, int W, int L, int H, const int32_t *box_set, int n_box,
                             uint64_t seed0) {
    for (int k = k0; k < k1; ++k) {
        uint8_t *row = pool + (size_t)k * T * 4;
        bpp_mt rng;
        bpp_npmt_seed(&rng, (uint32_t)(seed0 + (uint64_t)k));
        for (int t = 0; t < T - 1; ++t) {
            const int32_t *b = box_set + 3 * bpp_npmt_below(&rng, (uint32_t)n_box);   /* np.random.randint(0, len(box_set)) */
            row[4 * t] = (uint8_t)b[0];
            row[4 * t + 1] = (uint8_t)b[1];
            row[4 * t + 2] = (uint8_t)b[2];
            row[4 * t + 3] = 0;
        }
        row[4 * (T - 1)] = (uint8_t)W;
        row[4 * (T - 1) + 1] = (uint8_t)L;
        row[4 * (T - 1) + 2] = (uint8_t)H;
        row[4 * (T - 1) + 3] = 0;
    }
}

static int bpp_gen_rs_args_ok(int n, int T, int W, int L, int H, const int32_t *box_set, int n_box) {
    if (n <= 0 || T < 2 || W <= 0 || L <= 0 || H <= 0 || W > 255 || L > 255 || H > 255 || !box_set || n_box <= 0) return 0;
    for (int i = 0; i < 3 * n_box; ++i)
        if (box_set[i] < 1 || box_set[i] > 255) return 0;
    return 1;
}
