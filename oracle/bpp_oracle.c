/*
 * bpp_oracle.c -- TEST INFRASTRUCTURE, NOT THE PRODUCT.
 *
 * Plain-C, single-threaded CPU restatement of the reference environment step, used only as the
 * parity checker (tests/, __graft_entry__.smoke()) and as the `cpu_baseline` leg of bench.py.
 * The product path (libbpp_hip.so) never links, loads or calls this file.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against golden vectors
 * recorded from the unmodified reference Python (tests/golden/make_golden.py, run in the build
 * container where /root/reference exists) and tests/test_oracle_vs_reference.py re-runs the live
 * reference beside it when the reference tree is present.
 *
 * It deliberately follows the reference's arithmetic literally (float64 ratio compares, float64
 * reward), not the integer rewrites the HIP kernels use, so that agreement HIP == oracle also
 * validates those rewrites.  Paths below are relative to the reference root.
 */
#include "../include/bpp_abi.h"
#include "../include/bpp_gen.inl"

#include <math.h>
#include <stdio.h>
#include <string.h>

static __thread char g_err[256];

static int fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int bpp_abi_version(void) { return BPP_ABI_VERSION; }
const char *bpp_last_error(void) { return g_err; }
/* the oracle has no launch shapes: the knobs are accepted and remembered, nothing depends on them */
static bpp_knobs g_knobs = {0, 0, 1, 0, 0, 0, 0, 0, 1, {0, 0, 0}};
int bpp_get_knobs(bpp_knobs *out) {
    if (!out) return fail(BPP_E_BADARG, "bpp_get_knobs: NULL");
    *out = g_knobs;
    return 0;
}
int bpp_set_knobs(const bpp_knobs *k) {
    if (!k) return fail(BPP_E_BADARG, "bpp_set_knobs: NULL");
    g_knobs = *k;
    return 0;
}
int bpp_launch_info(int32_t E, int32_t W, int32_t L, int32_t H, int32_t rotation, int32_t out[6]) {
    (void)E; (void)W; (void)L; (void)H; (void)rotation;
    if (!out) return fail(BPP_E_BADARG, "bpp_launch_info: NULL");
    for (int k = 0; k < 6; ++k) out[k] = -1;   /* the oracle launches nothing */
    return 0;
}
int bpp_limits(int32_t out[2]) {
    if (!out) return fail(BPP_E_BADARG, "bpp_limits: NULL");
    out[0] = 1 << 20;
    out[1] = 255;
    return 0;
}

/* acktr/utils.py:8-35  check_box(plain, x, y, lx, ly, z, container_size) -- mask rule "U". */
static int check_box_utils(const int32_t *plain, int W, int L, int H, int x, int y, int lx, int ly, int z) {
    if (lx + x > W || ly + y > L) return -1;            /* :9-10 */
    if (lx < 0 || ly < 0) return -1;                    /* :11-12 */
    int max_h = plain[lx * L + ly];                     /* :14-15 np.max(rec) */
    for (int i = lx; i < lx + x; ++i)
        for (int j = ly; j < ly + y; ++j)
            if (plain[i * L + j] > max_h) max_h = plain[i * L + j];
    int max_area = 0;                                   /* :16 np.sum(rec == max_h) */
    for (int i = lx; i < lx + x; ++i)
        for (int j = ly; j < ly + y; ++j)
            max_area += (plain[i * L + j] == max_h);
    int area = x * y;                                   /* :17 */
    if (max_h + z > H) return -1;                       /* :20-21 */
    int LU = plain[lx * L + ly] == max_h;               /* :23-26 */
    int LD = plain[(lx + x - 1) * L + ly] == max_h;
    int RU = plain[lx * L + ly + y - 1] == max_h;
    int RD = plain[(lx + x - 1) * L + ly + y - 1] == max_h;
    double r = (double)max_area / (double)area;         /* Python true division */
    if (r > 0.95) return max_h;                         /* :28-29 */
    if (LU + LD + RU + RD == 3 && r > 0.85) return max_h; /* :30-31 */
    if (LU + LD + RU + RD == 4 && r > 0.50) return max_h; /* :32-33 */
    return -1;
}

/* envs/bpp0/space.py:111-144  Space.check_box -- placement rule "S". */
static int check_box_space(const int32_t *plain, int W, int L, int H, int x, int y, int lx, int ly, int z) {
    if (lx + x > W || ly + y > L) return -1;            /* :112-113 */
    if (lx < 0 || ly < 0) return -1;                    /* :114-115 */
    int r00 = plain[lx * L + ly];                       /* :117-121 */
    int r10 = plain[(lx + x - 1) * L + ly];
    int r01 = plain[lx * L + ly + y - 1];
    int r11 = plain[(lx + x - 1) * L + ly + y - 1];
    int rm = r00;                                       /* :122 */
    if (r10 > rm) rm = r10;
    if (r01 > rm) rm = r01;
    if (r11 > rm) rm = r11;
    int sc = (r00 == rm) + (r10 == rm) + (r01 == rm) + (r11 == rm); /* :123 */
    if (sc < 3) return -1;                              /* :124-125 */
    int max_h = r00;                                    /* :127 */
    for (int i = lx; i < lx + x; ++i)
        for (int j = ly; j < ly + y; ++j)
            if (plain[i * L + j] > max_h) max_h = plain[i * L + j];
    int max_area = 0;                                   /* :129 */
    for (int i = lx; i < lx + x; ++i)
        for (int j = ly; j < ly + y; ++j)
            max_area += (plain[i * L + j] == max_h);
    int area = x * y;                                   /* :130 */
    if (max_h + z > H) return -1;                       /* :134-135 */
    double r = (double)max_area / (double)area;
    if (r > 0.95) return max_h;                         /* :137-138 */
    if (rm == max_h && sc == 3 && r > 0.85) return max_h; /* :139-140 */
    if (rm == max_h && sc == 4 && r > 0.50) return max_h; /* :141-142 */
    return -1;
}

static int check_box(int rule, const int32_t *plain, int W, int L, int H, int x, int y, int lx, int ly, int z) {
    return rule == BPP_RULE_SPACE ? check_box_space(plain, W, L, H, x, y, lx, ly, z)
                                  : check_box_utils(plain, W, L, H, x, y, lx, ly, z);
}

/* acktr/utils.py:37-62 get_possible_position and :64-94 get_rotation_mask (rule U), or
 * envs/bpp0/bin3D.py:72-93 (rule S, rotation=0).  Writes float32 0/1 of length A*(1+rotation). */
static void build_mask(int rule, const int32_t *plain, int W, int L, int H, int x, int y, int z, int rotation,
                       float *mask) {
    int A = W * L, M = A * (1 + rotation), sum = 0;
    for (int k = 0; k < M; ++k) mask[k] = 0.0f;
    for (int i = 0; i < W - x + 1; ++i)                 /* utils.py:54-57 / :76-79 */
        for (int j = 0; j < L - y + 1; ++j)
            if (check_box(rule, plain, W, L, H, x, y, i, j, z) >= 0) {
                mask[i * L + j] = 1.0f;
                ++sum;
            }
    if (rotation)
        for (int i = 0; i < W - y + 1; ++i)             /* utils.py:81-84 */
            for (int j = 0; j < L - x + 1; ++j)
                if (check_box(rule, plain, W, L, H, y, x, i, j, z) >= 0) {
                    mask[A + i * L + j] = 1.0f;         /* :86 hstack */
                    ++sum;
                }
    if (sum == 0)                                       /* utils.py:59-60 / :91-92 */
        for (int k = 0; k < M; ++k) mask[k] = 1.0f;
}

static int check_batch(const bpp_batch *b) {
    if (!b || !b->seq_pool || !b->hmap || !b->state) return fail(BPP_E_BADARG, "bpp_batch: NULL pointer");
    if (b->num_envs <= 0 || b->W <= 0 || b->L <= 0 || b->H <= 0 || b->pool_size <= 0 || b->pool_len <= 0)
        return fail(BPP_E_BADARG, "bpp_batch: non-positive size");
    if (b->env_id_total < b->env_id_base + b->num_envs) return fail(BPP_E_BADARG, "bpp_batch: env_id_total too small");
    if (b->mask_rule != BPP_RULE_UTILS && b->mask_rule != BPP_RULE_SPACE) return fail(BPP_E_BADARG, "bpp_batch: bad mask_rule");
    if (b->H > 255) return fail(BPP_E_TOOLARGE, "bpp_batch: H > 255");
    if (b->pool_mode != BPP_POOL_STATIC && b->pool_mode != BPP_POOL_RING) return fail(BPP_E_BADARG, "bpp_batch: unknown pool_mode");
    if (b->pool_mode == BPP_POOL_RING && (b->pool_size % b->num_envs != 0 || b->pool_size / b->num_envs < 4))
        return fail(BPP_E_BADARG, "bpp_batch: a ring pool holds depth * num_envs rows, depth >= 4");
    if (b->pool_mode == BPP_POOL_RING && b->pool_len < 4) return fail(BPP_E_BADARG, "bpp_batch: ring rows need pool_len >= 4");
    return 0;
}

/* Pool row episode `episode` of local bin e plays (include/bpp_abi.h: bpp_batch.pool_mode). */
static int64_t pool_row(const bpp_batch *b, int e, int64_t episode) {
    if (b->pool_mode == BPP_POOL_RING) {
        int64_t depth = b->pool_size / b->num_envs;
        return (episode % depth) * b->num_envs + e;
    }
    return (b->env_id_base + e + episode * b->env_id_total) % b->pool_size;
}

/* Row of the episode after the one in row `seq` (the record's `seq` field is what a copied bin carries along, so a
 * copy keeps playing its source's sequences: copy.deepcopy(env) semantics, acktr/reorder.py:247). */
static int64_t next_row(const bpp_batch *b, int64_t seq) {
    int64_t stride = b->pool_mode == BPP_POOL_RING ? b->num_envs : b->env_id_total % b->pool_size;
    return (seq + stride) % b->pool_size;
}

/* BoxCreator.preview(1)[0] (envs/bpp0/binCreator.py:15-18): the item the bin is about to place.  It is read from the
 * state record (item_cur), which refresh_item_cache() below fills from the pooled sequence after every reset / step --
 * so it IS the pool entry at (seq, cursor) unless a caller overwrote it in between (BppVecEnv.set_current_items: the
 * reorder search of acktr/reorder.py:181-215 plays previewed items in another order), exactly what the kernels play. */
static void next_box(const bpp_batch *b, int e, const bpp_env_state *s, int item[3]) {
    (void)b;
    (void)e;
    item[0] = (int)(s->item_cur & 255u);
    item[1] = (int)((s->item_cur >> 8) & 255u);
    item[2] = (int)((s->item_cur >> 16) & 255u);
}

/* Keep the pool-entry cache of the state record coherent (see include/bpp_abi.h). */
static uint32_t pool_entry(const bpp_batch *b, int64_t seq, int c) {
    /* item c of row seq.  The rows of a stream's ring start with two look-ahead entries (include/bpp_abi.h) that this
       library writes but never reads: it looks the following rows up directly -- which is what checks the kernels'
       use of those entries. */
    const int hdr = b->pool_mode == BPP_POOL_RING ? 2 : 0;
    if (c > b->pool_len - 1 - hdr) c = b->pool_len - 1 - hdr;
    c += hdr;
    const uint8_t *p = b->seq_pool + ((size_t)seq * b->pool_len + c) * 4;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}
static void refresh_item_cache(const bpp_batch *b, int e, bpp_env_state *s) {
    (void)e;
    int64_t seq_n = next_row(b, s->seq);
    s->item_cur = pool_entry(b, s->seq, s->cursor);
    s->item_next = pool_entry(b, s->seq, s->cursor + 1);
    s->item_reset = pool_entry(b, seq_n, 0);
}

/* PackingGame.cur_observation (envs/bpp0/bin3D.py:61-66) as float32 (shmem_vec_env.py:42-43) + mask. */
static void write_obs_mask(const bpp_batch *b, int e, const int item[3], const bpp_step_out *out) {
    int A = b->W * b->L, M = A * (1 + b->rotation);
    int32_t plain[A];
    for (int k = 0; k < A; ++k) plain[k] = b->hmap[(size_t)e * A + k];
    float *o = out->obs + (size_t)e * 4 * A;
    for (int k = 0; k < A; ++k) {
        o[k] = (float)plain[k];
        o[A + k] = (float)item[0];                      /* bin3D.py:49-53 */
        o[2 * A + k] = (float)item[1];
        o[3 * A + k] = (float)item[2];
    }
    if (out->mask)
        build_mask(b->mask_rule, plain, b->W, b->L, b->H, item[0], item[1], item[2], b->rotation,
                   out->mask + (size_t)e * M);
}

/* PackingGame.reset (bin3D.py:55-59) + Monitor.reset_state (bench/monitor.py:45-49). */
static void reset_bin(const bpp_batch *b, int e, bpp_env_state *s, int first) {
    int A = b->W * b->L;
    memset(b->hmap + (size_t)e * A, 0, (size_t)A); /* space.py:22 */
    s->cursor = 0;
    s->n_boxes = 0;
    s->vol_sum = 0;
    s->ep_ret = 0.0;
    s->ep_len = 0;
    s->hmax = 0;
    s->seq = (int32_t)(first ? pool_row(b, e, 0) : next_row(b, s->seq));
}

int bpp_reset(const bpp_batch *b, int32_t mode, const bpp_step_out *out, void *stream) {
    (void)stream;
    int rc = check_batch(b);
    if (rc) return rc;
    if (!out || !out->obs) return fail(BPP_E_BADARG, "bpp_reset: NULL obs");
    if (mode != BPP_RESET_INIT && mode != BPP_RESET_ADVANCE) return fail(BPP_E_BADARG, "bpp_reset: bad mode");
    for (int e = 0; e < b->num_envs; ++e) {
        bpp_env_state *s = b->state + e;
        s->episode = mode == BPP_RESET_INIT ? 0 : s->episode + 1;
        reset_bin(b, e, s, mode == BPP_RESET_INIT);
        int item[3];
        refresh_item_cache(b, e, s);
        next_box(b, e, s, item);
        write_obs_mask(b, e, item, out);
    }
    return 0;
}

int bpp_step(const bpp_batch *b, const int64_t *actions, const bpp_step_out *out, void *stream) {
    (void)stream;
    int rc = check_batch(b);
    if (rc) return rc;
    if (!actions || !out || !out->obs || !out->reward || !out->done || !out->counter || !out->ratio ||
        !out->ep_ret || !out->ep_len)
        return fail(BPP_E_BADARG, "bpp_step: NULL pointer");
    if ((out->host_reward == NULL) != (out->host_done == NULL))
        return fail(BPP_E_BADARG, "bpp_step_out: host_reward and host_done go together");
    const int W = b->W, L = b->L, H = b->H, A = W * L;
    for (int e = 0; e < b->num_envs; ++e) {
        bpp_env_state *s = b->state + e;
        uint8_t *bytes = b->hmap + (size_t)e * A;      /* state is kept as bytes; work on an int32 copy */
        int32_t plain[A];
        for (int k = 0; k < A; ++k) plain[k] = bytes[k];
        int item[3];
        next_box(b, e, s, item);
        if (actions[e] == BPP_ACTION_NOOP) {            /* include/bpp_abi.h: the bin is left alone */
            out->counter[e] = s->n_boxes;
            out->ratio[e] = (double)s->vol_sum / ((double)W * (double)L * (double)H);
            out->ep_ret[e] = s->ep_ret;
            out->ep_len[e] = s->ep_len;
            out->reward[e] = 0.0f;
            out->done[e] = 0;
            if (out->host_reward) out->host_reward[e] = 0.0f, out->host_done[e] = 0;
            write_obs_mask(b, e, item, out);
            continue;
        }
        /* bin3D.py:96-105 */
        int64_t idx = actions[e];
        int flag = 0;
        if (idx > A && b->rotation) {                   /* strict '>' : quirk A.6-1.  With rotation off
                                                           the reference asserts; such an index is outside
                                                           the action space and is left to land out of
                                                           bounds below (lx >= W). */
            idx -= A;
            flag = 1;
        }
        /* space.py:164-172 */
        int x = flag ? item[1] : item[0];
        int y = flag ? item[0] : item[1];
        int z = item[2];
        int new_h = -1;
        /* space.py:153-156 idx_to_position.  A negative idx floors to lx < 0 in Python -> -1;
           idx >= (W+1)*L is out of bounds whatever the item -> -1 (also keeps lx in int range). */
        if (idx >= 0 && idx < (int64_t)(W + 1) * L)
            new_h = check_box_space(plain, W, L, H, x, y, (int)(idx / L), (int)(idx % L), z);
        double binvol = (double)W * (double)L * (double)H;
        double reward;
        int done;
        if (new_h != -1) {
            /* space.py:175-178 + update_height_graph :36-46 */
            int lx = (int)(idx / L), ly = (int)(idx % L);
            int max_h = plain[lx * L + ly];
            for (int i = lx; i < lx + x; ++i)
                for (int j = ly; j < ly + y; ++j)
                    if (plain[i * L + j] > max_h) max_h = plain[i * L + j];
            if (new_h + z > max_h) max_h = new_h + z;
            for (int i = lx; i < lx + x; ++i)
                for (int j = ly; j < ly + y; ++j) bytes[i * L + j] = (uint8_t)max_h;
            s->n_boxes += 1;
            s->vol_sum += x * y * z;
            if ((uint32_t)max_h > s->hmax) s->hmax = (uint32_t)max_h;   /* include/bpp_abi.h: highest cell of the bin */
            /* bin3D.py:44-46,114,121: float64 (vol / binvol) * 10 */
            reward = ((double)(item[0] * item[1] * item[2]) / binvol) * 10.0;
            s->cursor += 1;                             /* bin3D.py:116-117 drop_box + generate_box_size */
            done = 0;
        } else {
            reward = 0.0;                               /* bin3D.py:108-112 */
            done = 1;
        }
        /* info: bin3D.py:111,123-125 ; get_ratio space.py:146-151 */
        out->counter[e] = s->n_boxes;
        out->ratio[e] = (double)s->vol_sum / binvol;
        /* bench/monitor.py:58-64 */
        s->ep_ret += reward;
        s->ep_len += 1;
        out->ep_ret[e] = s->ep_ret;
        out->ep_len[e] = s->ep_len;
        out->reward[e] = (float)reward;                 /* acktr/envs.py:192 .float() */
        out->done[e] = (uint8_t)done;
        if (out->host_reward) out->host_reward[e] = (float)reward, out->host_done[e] = (uint8_t)done;
        if (done && b->ep_acc) {                        /* main.py:159-162: the bin's own accumulator row */
            double *a = b->ep_acc + 4 * (size_t)e;
            a[0] += s->ep_ret;
            a[1] += out->ratio[e];
            a[2] += (double)s->ep_len;
            a[3] += 1.0;
        }
        if (done) {                                     /* shmem_vec_env.py:128-129 */
            s->episode += 1;
            reset_bin(b, e, s, 0);
        }
        refresh_item_cache(b, e, s);                    /* from the pool at the new (seq, cursor) */
        next_box(b, e, s, item);
        write_obs_mask(b, e, item, out);
    }
    if (out->next_action) {
        if (!out->mask) return fail(BPP_E_BADARG, "bpp_step: next_action needs mask");
        return bpp_sample_feasible(out->mask, out->next_action, b->num_envs, A * (1 + b->rotation), b->env_id_base,
                                   out->sample_seed, out->sample_step, stream);
    }
    return 0;
}

int bpp_mask_from_obs(const float *obs, float *mask, int32_t E, int32_t W, int32_t L, int32_t H,
                      int32_t rotation, int32_t rule, void *stream) {
    (void)stream;
    if (!obs || !mask) return fail(BPP_E_BADARG, "bpp_mask_from_obs: NULL pointer");
    if (E <= 0 || W <= 0 || L <= 0 || H <= 0) return fail(BPP_E_BADARG, "bpp_mask_from_obs: non-positive size");
    if (rule != BPP_RULE_UTILS && rule != BPP_RULE_SPACE) return fail(BPP_E_BADARG, "bpp_mask_from_obs: bad rule");
    int A = W * L, M = A * (1 + rotation);
    int32_t plain[A];
    for (int e = 0; e < E; ++e) {
        const float *o = obs + (size_t)e * 4 * A;
        for (int k = 0; k < A; ++k) plain[k] = (int32_t)o[k];
        /* acktr/utils.py:43-45: int(box_info[k][0]) */
        build_mask(rule, plain, W, L, H, (int)o[A], (int)o[2 * A], (int)o[3 * A], rotation, mask + (size_t)e * M);
    }
    return 0;
}

int bpp_mask_from_hmap(const int32_t *hmap, const int32_t *items, float *mask, int32_t E, int32_t W,
                       int32_t L, int32_t H, int32_t rotation, int32_t rule, void *stream) {
    (void)stream;
    if (!hmap || !items || !mask) return fail(BPP_E_BADARG, "bpp_mask_from_hmap: NULL pointer");
    if (E <= 0 || W <= 0 || L <= 0 || H <= 0) return fail(BPP_E_BADARG, "bpp_mask_from_hmap: non-positive size");
    if (rule != BPP_RULE_UTILS && rule != BPP_RULE_SPACE) return fail(BPP_E_BADARG, "bpp_mask_from_hmap: bad rule");
    int A = W * L, M = A * (1 + rotation);
    for (int e = 0; e < E; ++e)
        build_mask(rule, hmap + (size_t)e * A, W, L, H, items[3 * e], items[3 * e + 1], items[3 * e + 2], rotation,
                   mask + (size_t)e * M);
    return 0;
}

/* Counter-based RNG of the benchmark/test action sources (no reference counterpart): a 32-bit
 * multiply-xorshift hash of (seed, global bin id, step). */
static uint32_t mix32(uint64_t seed, uint64_t gid, uint64_t step) {
    uint32_t h = ((uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u)) ^
                 (((uint32_t)step + (uint32_t)(step >> 32) * 0xC2B2AE3Du) * 0x27D4EB2Fu);
    h ^= (uint32_t)gid * 0x85EBCA77u;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

int bpp_sample_feasible(const float *mask, int64_t *actions, int32_t E, int32_t M, int64_t env_id_base,
                        uint64_t seed, uint64_t step, void *stream) {
    (void)stream;
    if (!mask || !actions) return fail(BPP_E_BADARG, "bpp_sample_feasible: NULL pointer");
    if (E <= 0 || M <= 0) return fail(BPP_E_BADARG, "bpp_sample_feasible: non-positive size");
    for (int e = 0; e < E; ++e) {
        const float *m = mask + (size_t)e * M;
        int cnt = 0;
        for (int k = 0; k < M; ++k) cnt += (m[k] != 0.0f);
        if (cnt == 0) {
            actions[e] = 0;
            continue;
        }
        uint64_t pick = ((uint64_t)mix32(seed, (uint64_t)(env_id_base + e), step) * (uint64_t)cnt) >> 32;
        int64_t a = 0;
        for (int k = 0; k < M; ++k)
            if (m[k] != 0.0f) {
                if (pick == 0) {
                    a = k;
                    break;
                }
                --pick;
            }
        actions[e] = a;
    }
    return 0;
}

/* include/bpp_abi.h: for d = 512 .. 1: partial[r] += partial[r + d] (r < d); acc[k] += partial[0] */
static void reduce_tree(double part[4][BPP_REDUCE_LANES], double *acc) {
    for (int d = BPP_REDUCE_LANES / 2; d > 0; d >>= 1)
        for (int r = 0; r < d; ++r)
            for (int k = 0; k < 4; ++k) part[k][r] += part[k][r + d];
    for (int k = 0; k < 4; ++k) acc[k] += part[k][0];
}

int bpp_episode_stats(const uint8_t *done, const double *ep_ret, const double *ratio, const int32_t *ep_len,
                      int32_t E, double *acc, void *stream) {
    (void)stream;
    if (!done || !ep_ret || !ratio || !ep_len || !acc) return fail(BPP_E_BADARG, "bpp_episode_stats: NULL pointer");
    if (E <= 0) return fail(BPP_E_BADARG, "bpp_episode_stats: non-positive size");
    /* main.py:159-162; summation order as include/bpp_abi.h states it */
    static double part[4][BPP_REDUCE_LANES];
    for (int r = 0; r < BPP_REDUCE_LANES; ++r) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        for (int e = r; e < E; e += BPP_REDUCE_LANES)
            if (done[e]) {
                s0 += ep_ret[e];
                s1 += ratio[e];
                s2 += (double)ep_len[e];
                s3 += 1.0;
            }
        part[0][r] = s0, part[1][r] = s1, part[2][r] = s2, part[3][r] = s3;
    }
    reduce_tree(part, acc);
    return 0;
}

int bpp_episode_acc_reduce(double *ep_acc, int32_t E, double *acc, int32_t clear, void *scratch, void *stream) {
    (void)scratch;
    (void)stream;
    if (!ep_acc || !acc) return fail(BPP_E_BADARG, "bpp_episode_acc_reduce: NULL pointer");
    if (E <= 0) return fail(BPP_E_BADARG, "bpp_episode_acc_reduce: non-positive size");
    static double part[4][BPP_REDUCE_LANES];
    for (int r = 0; r < BPP_REDUCE_LANES; ++r) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int e = r; e < E; e += BPP_REDUCE_LANES)
            for (int k = 0; k < 4; ++k) s[k] += ep_acc[4 * (size_t)e + k];
        for (int k = 0; k < 4; ++k) part[k][r] = s[k];
    }
    reduce_tree(part, acc);
    if (clear) memset(ep_acc, 0, (size_t)E * 4 * sizeof(double));
    return 0;
}

int bpp_gather_finished(const uint8_t *done, const double *ep_ret, const double *ratio, const int32_t *ep_len,
                        const int32_t *counter, int32_t E, void *dev, void *host, int32_t n, void *stream) {
    (void)stream;
    if (!done || !ep_ret || !ratio || !ep_len || !counter || !host) return fail(BPP_E_BADARG, "bpp_gather_finished: NULL pointer");
    if (E <= 0 || n < BPP_GATHER_ENQUEUE_ONLY || n > E) return fail(BPP_E_BADARG, "bpp_gather_finished: bad size");
    int eager = n == BPP_GATHER_ENQUEUE_ONLY;       /* arrays laid out for E entries, no count check (include/bpp_abi.h) */
    if (eager && dev) return fail(BPP_E_BADARG, "bpp_gather_finished: BPP_GATHER_ENQUEUE_ONLY writes into mapped host memory (dev must be NULL)");
    if (eager) n = E;
    if (!dev) dev = host;
    unsigned char *out = (unsigned char *)dev;
    double *o_ret = (double *)(out + 32), *o_ratio = o_ret + n;
    int32_t *o_len = (int32_t *)(o_ratio + n), *o_cnt = o_len + n, *o_bin = o_cnt + n;
    int k = 0;
    for (int e = 0; e < E; ++e)
        if (done[e]) {
            if (k < n) o_ret[k] = ep_ret[e], o_ratio[k] = ratio[e], o_len[k] = ep_len[e], o_cnt[k] = counter[e], o_bin[k] = e;
            ++k;
        }
    memset(out, 0, 32);
    *(int32_t *)out = k;
    if (dev != host) memmove(host, dev, (size_t)BPP_FINISHED_BYTES(n));
    if (!eager && k != n) return fail(BPP_E_BADARG, "bpp_gather_finished: n is not the number of finished bins of this step");
    return 0;
}

int bpp_fetch_to_host(const void *device_src, void *host_dst, int64_t nbytes, void *stream) {
    (void)stream;
    if (!device_src || !host_dst || nbytes <= 0) return fail(BPP_E_BADARG, "bpp_fetch_to_host: NULL pointer / non-positive size");
    memmove(host_dst, device_src, (size_t)nbytes);      /* host pointers on both sides here */
    return 0;
}

int bpp_wait(void *stream) {
    (void)stream;
    return 0;
}

int bpp_mark(void *host_flag, uint32_t value, void *stream) {       /* no queue here: everything before is complete */
    (void)stream;
    if (!host_flag) return fail(BPP_E_BADARG, "bpp_mark: NULL flag");
    *(uint32_t *)host_flag = value;
    return 0;
}

int bpp_wait_mark(const void *host_flag, uint32_t value, void *stream) {
    (void)stream;
    if (!host_flag) return fail(BPP_E_BADARG, "bpp_wait_mark: NULL flag");
    return *(const uint32_t *)host_flag == value ? 0 : fail(BPP_E_BADARG, "bpp_wait_mark: the flag does not hold the value (no bpp_mark?)");
}

int bpp_step_dropin(const bpp_batch *b, const int64_t *actions, const bpp_step_out *out, void *fin_host, void *host_flag,
                    uint32_t value, void *stream) {      /* include/bpp_abi.h: the three calls one after the other */
    int rc = bpp_step(b, actions, out, stream);
    if (rc == 0 && fin_host)
        rc = bpp_gather_finished(out->done, out->ep_ret, out->ratio, out->ep_len, out->counter, b->num_envs, NULL, fin_host,
                                 BPP_GATHER_ENQUEUE_ONLY, stream);
    if (rc == 0 && host_flag) rc = bpp_mark(host_flag, value, stream);
    return rc;
}

int bpp_epsilon_override(int64_t *actions, int32_t E, int32_t M, int64_t env_id_base, uint64_t seed, uint64_t step, uint32_t eps_q24,
                         void *stream) {
    (void)stream;
    if (!actions) return fail(BPP_E_BADARG, "bpp_epsilon_override: NULL pointer");
    if (E <= 0 || M <= 0 || eps_q24 > (1u << 24)) return fail(BPP_E_BADARG, "bpp_epsilon_override: bad size / eps_q24 > 2^24");
    for (int e = 0; e < E; ++e) {      /* include/bpp_abi.h: coin and pick are two independent hashes of (seed, bin, step) */
        uint64_t gid = (uint64_t)(env_id_base + e);
        if ((mix32(seed ^ BPP_EPS_KEY_COIN, gid, step) >> 8) < eps_q24)
            actions[e] = (int64_t)(((uint64_t)mix32(seed ^ BPP_EPS_KEY_PICK, gid, step) * (uint64_t)M) >> 32);
    }
    return 0;
}

int bpp_rollout_uniform_sets(const bpp_batch *b, const bpp_step_out *outs, int32_t nsets, const float *first_mask,
                             int64_t *actions, uint64_t seed, uint64_t step0, int32_t nsteps, int32_t flags, void *stream) {
    if (!b || !outs || !actions || nsets < 1) return fail(BPP_E_BADARG, "bpp_rollout_uniform_sets: NULL pointer / no output set");
    if (nsteps < 0) return fail(BPP_E_BADARG, "bpp_rollout_uniform_sets: negative nsteps");
    int M = b->W * b->L * (1 + b->rotation), rc = 0;
    uint32_t eps = BPP_ROLLOUT_EPS_OF(flags);
    if (nsteps == 0) return 0;
    if (!(flags & BPP_ROLLOUT_CONTINUE)) {
        if (!first_mask) return fail(BPP_E_BADARG, "bpp_rollout_uniform_sets: first_mask needed without BPP_ROLLOUT_CONTINUE");
        rc = bpp_sample_feasible(first_mask, actions, b->num_envs, M, b->env_id_base, seed, step0, stream);
        if (rc == 0 && eps) rc = bpp_epsilon_override(actions, b->num_envs, M, b->env_id_base, seed, step0, eps, stream);
    }
    for (int t = 0; rc == 0 && t < nsteps; ++t) {
        bpp_step_out o = outs[t % nsets];
        if (!o.mask) return fail(BPP_E_BADARG, "bpp_rollout_uniform_sets: every output set needs a mask");
        o.next_action = NULL;
        rc = bpp_step(b, actions, &o, stream);
        if (rc == 0)
            rc = bpp_sample_feasible(o.mask, actions, b->num_envs, M, b->env_id_base, seed, step0 + (uint64_t)t + 1, stream);
        if (rc == 0 && eps)
            rc = bpp_epsilon_override(actions, b->num_envs, M, b->env_id_base, seed, step0 + (uint64_t)t + 1, eps, stream);
    }
    return rc;
}

int bpp_rollout_uniform(const bpp_batch *b, const bpp_step_out *out, int64_t *actions, uint64_t seed,
                        uint64_t step0, int32_t nsteps, void *stream) {
    if (!b || !out || !out->mask || !actions) return fail(BPP_E_BADARG, "bpp_rollout_uniform: NULL pointer");
    if (nsteps < 0) return fail(BPP_E_BADARG, "bpp_rollout_uniform: negative nsteps");
    int M = b->W * b->L * (1 + b->rotation);
    for (int t = 0; t < nsteps; ++t) {
        int rc = bpp_sample_feasible(out->mask, actions, b->num_envs, M, b->env_id_base, seed, step0 + (uint64_t)t, stream);
        if (rc) return rc;
        bpp_step_out o = *out;
        o.next_action = NULL;
        rc = bpp_step(b, actions, &o, stream);
        if (rc) return rc;
    }
    return 0;
}

/* acktr/distributions.py:71-84 + torch.distributions.Categorical(probs=...) in float32, sequential sums. */
int bpp_masked_act(const float *logits, const float *mask, int64_t *action, float *log_prob, int32_t E, int32_t M,
                   int64_t env_id_base, uint64_t seed, uint64_t step, int32_t deterministic, void *stream) {
    (void)stream;
    if (!logits || !mask || !action) return fail(BPP_E_BADARG, "bpp_masked_act: NULL pointer");
    if (E <= 0 || M <= 0) return fail(BPP_E_BADARG, "bpp_masked_act: non-positive size");
    for (int e = 0; e < E; ++e) {
        const float *x = logits + (size_t)e * M, *m = mask + (size_t)e * M;
        float mx = -INFINITY, sum = 0.0f, tot = 0.0f;
        for (int k = 0; k < M; ++k) {
            float z = x[k] - (1.0f - m[k]) * 14.0f;
            if (z > mx) mx = z;
        }
        for (int k = 0; k < M; ++k) sum += expf(x[k] - (1.0f - m[k]) * 14.0f - mx);
        for (int k = 0; k < M; ++k) tot += expf(x[k] - (1.0f - m[k]) * 14.0f - mx) / sum + 1e-5f;
        int a = 0;
        if (deterministic) {
            float best = -1.0f;
            for (int k = 0; k < M; ++k) {
                float pk = expf(x[k] - (1.0f - m[k]) * 14.0f - mx) / sum + 1e-5f;
                if (pk > best) { best = pk; a = k; }
            }
        } else {
            float u = (float)(mix32(seed, (uint64_t)(env_id_base + e), step) >> 8) * (1.0f / 16777216.0f);
            float target = u * tot, c = 0.0f;
            a = M - 1;
            for (int k = 0; k < M; ++k) {
                c += expf(x[k] - (1.0f - m[k]) * 14.0f - mx) / sum + 1e-5f;
                if (c > target) { a = k; break; }
            }
        }
        action[e] = a;
        if (log_prob) {
            float q = (expf(x[a] - (1.0f - m[a]) * 14.0f - mx) / sum + 1e-5f) / tot;
            const float eps = 1.1920928955078125e-7f;
            if (q < eps) q = eps;
            if (q > 1.0f - eps) q = 1.0f - eps;
            log_prob[e] = logf(q);
        }
    }
    return 0;
}

int bpp_masked_act_counter(const float *logits, const float *mask, int64_t *action, float *log_prob, int32_t E, int32_t M,
                           int64_t env_id_base, const uint64_t *seed_step, int32_t deterministic, void *stream) {
    if (!seed_step) return fail(BPP_E_BADARG, "bpp_masked_act_counter: NULL seed_step");     /* include/bpp_abi.h: (seed, step) from memory */
    return bpp_masked_act(logits, mask, action, log_prob, E, M, env_id_base, seed_step[0], seed_step[1], deterministic, stream);
}

/* Training half: acktr/distributions.py:71-101 as consumed by Policy.evaluate_actions (acktr/model.py:90-96):
 * Categorical(probs = softmax(x - 14 (1 - m)) + 1e-5): log_prob of the taken action, entropy, and the row sum of
 * bx = softmax(x) * (1 - m).  float32, sequential sums. */
typedef struct { float mq, ma, sq, sa, tot; } row_stats;
static row_stats masked_row_stats(const float *x, const float *m, int M) {
    row_stats r = {-INFINITY, -INFINITY, 0.0f, 0.0f, 0.0f};
    for (int k = 0; k < M; ++k) {
        float z = x[k] - (1.0f - m[k]) * 14.0f;
        if (z > r.mq) r.mq = z;
        if (x[k] > r.ma) r.ma = x[k];
    }
    for (int k = 0; k < M; ++k) {
        r.sq += expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq);
        r.sa += expf(x[k] - r.ma);
    }
    for (int k = 0; k < M; ++k) r.tot += expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq) / r.sq + 1e-5f;
    return r;
}
static const float kProbEps = 1.1920928955078125e-7f;
static float clampp(float p) { return p < kProbEps ? kProbEps : (p > 1.0f - kProbEps ? 1.0f - kProbEps : p); }

int bpp_masked_evaluate(const float *logits, const float *mask, const int64_t *action, float *log_prob, float *entropy,
                        float *bad_prob, int32_t E, int32_t M, void *stream) {
    (void)stream;
    if (!logits || !mask || !action || !log_prob || !entropy || !bad_prob) return fail(BPP_E_BADARG, "bpp_masked_evaluate: NULL pointer");
    if (E <= 0 || M <= 0) return fail(BPP_E_BADARG, "bpp_masked_evaluate: non-positive size");
    for (int e = 0; e < E; ++e) {
        const float *x = logits + (size_t)e * M, *m = mask + (size_t)e * M;
        row_stats r = masked_row_stats(x, m, M);
        float h = 0.0f, b = 0.0f;
        for (int k = 0; k < M; ++k) {
            float p = (expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq) / r.sq + 1e-5f) / r.tot;
            h -= p * logf(clampp(p));
            b += expf(x[k] - r.ma) / r.sa * (1.0f - m[k]);
        }
        int64_t a = action[e];
        float pa = (a >= 0 && a < M) ? (expf(x[a] - (1.0f - m[a]) * 14.0f - r.mq) / r.sq + 1e-5f) / r.tot : kProbEps;
        log_prob[e] = logf(clampp(pa));
        entropy[e] = h;
        bad_prob[e] = b;
    }
    return 0;
}

/* Chain rule through p = lx / sum(lx), lx = q + 1e-5, q = softmax(z): with h = dLoss/dp, c = sum p h,
 * u = (h - c) / tot, v = sum q u:  dLoss/dx = q (u - v)  +  g_bad * a ((1 - m) - bad),  a = softmax(x). */
int bpp_masked_evaluate_backward(const float *logits, const float *mask, const int64_t *action, const float *g_log_prob,
                                 const float *g_entropy, const float *g_bad_prob, float *grad_logits, int32_t E, int32_t M,
                                 void *stream) {
    (void)stream;
    if (!logits || !mask || !action || !g_log_prob || !g_entropy || !g_bad_prob || !grad_logits)
        return fail(BPP_E_BADARG, "bpp_masked_evaluate_backward: NULL pointer");
    if (E <= 0 || M <= 0) return fail(BPP_E_BADARG, "bpp_masked_evaluate_backward: non-positive size");
    for (int e = 0; e < E; ++e) {
        const float *x = logits + (size_t)e * M, *m = mask + (size_t)e * M;
        float *g = grad_logits + (size_t)e * M;
        row_stats r = masked_row_stats(x, m, M);
        const int64_t a = action[e];
        float c = 0.0f, b = 0.0f, v = 0.0f;
        for (int pass = 0; pass < 3; ++pass)
            for (int k = 0; k < M; ++k) {
                float q = expf(x[k] - (1.0f - m[k]) * 14.0f - r.mq) / r.sq;
                float p = (q + 1e-5f) / r.tot, pc = clampp(p);
                int inside = p > kProbEps && p < 1.0f - kProbEps;
                float h = -g_entropy[e] * (logf(pc) + (inside ? p / pc : 0.0f));
                if (k == a) h += inside ? g_log_prob[e] / pc : 0.0f;
                float av = expf(x[k] - r.ma) / r.sa;
                if (pass == 0) {
                    c += p * h;
                    b += av * (1.0f - m[k]);
                } else if (pass == 1) {
                    v += q * (h - c) / r.tot;
                } else {
                    g[k] = q * ((h - c) / r.tot - v) + g_bad_prob[e] * av * ((1.0f - m[k]) - b);
                }
            }
    }
    return 0;
}

int bpp_gen_cut2(uint8_t *pool, int32_t *lengths, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H,
                 int32_t bound_lo, int32_t bound_hi, uint64_t seed0, int32_t threads) {
    (void)threads;
    if (!pool || !bpp_gen_cut2_args_ok(n, T, W, L, H, bound_lo, bound_hi))
        return fail(BPP_E_BADARG, "bpp_gen_cut2: bad argument");
    return bpp_gen_cut2_range(pool, lengths, 0, n, T, W, L, H, bound_lo, bound_hi, seed0)
               ? fail(BPP_E_TOOLARGE, "bpp_gen_cut2: a sequence does not fit in T-1 entries") : 0;
}

int bpp_gen_cut1(uint8_t *pool, int32_t *lengths, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H,
                 const int32_t box_range[6], int32_t rotation, uint64_t seed0, int32_t threads) {
    (void)threads;
    if (!pool || !bpp_gen_cut1_args_ok(n, T, W, L, H, box_range) || seed0 + (uint64_t)n > (1ull << 32))
        return fail(BPP_E_BADARG, "bpp_gen_cut1: bad argument");
    int st = bpp_gen_cut1_range(pool, lengths, 0, n, T, W, L, H, box_range, rotation != 0, seed0);
    if (st == 2) return fail(BPP_E_BADARG, "bpp_gen_cut1: a piece fell below the lower bound");
    return st ? fail(BPP_E_TOOLARGE, "bpp_gen_cut1: a sequence does not fit in T-1 entries") : 0;
}

int bpp_gen_rs(uint8_t *pool, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H, const int32_t *box_set, int32_t n_box,
               uint64_t seed0, int32_t threads) {
    (void)threads;
    if (!pool || !bpp_gen_rs_args_ok(n, T, W, L, H, box_set, n_box) || seed0 + (uint64_t)n > (1ull << 32))
        return fail(BPP_E_BADARG, "bpp_gen_rs: bad argument");
    bpp_gen_rs_range(pool, 0, n, T, W, L, H, box_set, n_box, seed0);
    return 0;
}

/* ---- endless CUT-2 supply (include/bpp_abi.h: bpp_stream), host version: one bpp_mt per bin in `mt` (the buffer is
 * opaque, 625 words per bin either way), sequences drawn one after the other with include/bpp_gen.inl ---------- */
static int check_stream(const bpp_stream *s) {
    if (!s || !s->ring || !s->mt || !s->work || !s->gen_next || !s->state) return fail(BPP_E_BADARG, "bpp_stream: NULL pointer");
    if (s->num_envs <= 0 || s->depth < 4 || s->pool_len < 4 || s->env_id_base < 0) return fail(BPP_E_BADARG, "bpp_stream: bad size");
    if (!bpp_gen_cut2_args_ok(1, s->pool_len, s->W, s->L, s->H, s->bound_lo, s->bound_hi)) return fail(BPP_E_BADARG, "bpp_stream: bad bounds");
    if (s->rng != BPP_STREAM_RNG_MT19937 && s->rng != BPP_STREAM_RNG_COUNTER) return fail(BPP_E_BADARG, "bpp_stream: unknown rng");
    return 0;
}

int bpp_stream_sizes(const bpp_stream *s, int64_t out[2]) {
    if (!s || !out) return fail(BPP_E_BADARG, "bpp_stream_sizes: NULL pointer");
    if (s->num_envs <= 0) return fail(BPP_E_BADARG, "bpp_stream_sizes: bad size");
    /* one bpp_mt per bin, or (counter generator) four words: the bin's 64-bit stream id */
    out[0] = (int64_t)(s->rng == BPP_STREAM_RNG_COUNTER ? 4 : sizeof(bpp_mt) / 4) * s->num_envs;
    out[1] = 16;                                            /* no scratch needed on the host */
    return 0;
}

int bpp_stream_init(const bpp_stream *s, void *stream) {
    (void)stream;
    int rc = check_stream(s);
    if (rc) return rc;
    bpp_mt *rngs = (bpp_mt *)s->mt;
    for (int e = 0; e < s->num_envs; ++e) {
        if (s->rng == BPP_STREAM_RNG_COUNTER) {
            const uint64_t sid = (uint64_t)(s->env_id_base + e);
            s->mt[4 * e] = (uint32_t)sid, s->mt[4 * e + 1] = (uint32_t)(sid >> 32), s->mt[4 * e + 2] = 0, s->mt[4 * e + 3] = 0;
        } else {
            bpp_mt_seed(&rngs[e], s->seed0 + (uint64_t)(s->env_id_base + e));
        }
        s->gen_next[e] = 0;
    }
    return 0;
}

int bpp_stream_refill(const bpp_stream *s, void *stream) {
    (void)stream;
    int rc = check_stream(s);
    if (rc) return rc;
    bpp_mt *rngs = (bpp_mt *)s->mt;
    const int E = s->num_envs, T = s->pool_len, D = s->depth;
    for (int e = 0; e < E; ++e) {
        while (s->gen_next[e] < s->state[e].episode + D) {
            const int g = s->gen_next[e];
            uint8_t *base = s->ring + ((size_t)(g % D) * E + e) * T * 4;
            uint8_t *row = base + 2 * 4;                  /* items behind the two look-ahead entries */
            for (int t = 0; t < T - 2; ++t) {
                row[4 * t] = (uint8_t)s->W;
                row[4 * t + 1] = (uint8_t)s->L;
                row[4 * t + 2] = (uint8_t)s->H;
                row[4 * t + 3] = 0;
            }
            int n;
            if (s->rng == BPP_STREAM_RNG_COUNTER)       /* a pure function of (seed0, stream id, episode) */
                n = bpp_cut2_counter(s->seed0, (uint64_t)s->mt[4 * e] | ((uint64_t)s->mt[4 * e + 1] << 32), (uint32_t)g,
                                     s->W, s->L, s->H, s->bound_lo, s->bound_hi, row, T - 3);
            else
                n = bpp_cut2_from_stream(&rngs[e], s->W, s->L, s->H, s->bound_lo, s->bound_hi, row, T - 3);
            if (n > T - 3 && s->overflow) s->overflow[0] += 1;
            /* this row's item 1 -> entry 0 of the row before, its item 0 -> entry 1 of the row two before */
            if (g >= 1) memcpy(s->ring + ((size_t)((g - 1) % D) * E + e) * T * 4, row + 4, 4);
            if (g >= 2) memcpy(s->ring + ((size_t)((g - 2) % D) * E + e) * T * 4 + 4, row, 4);
            s->gen_next[e] += 1;
        }
    }
    return 0;
}

int bpp_side_create(void **side) {      /* host twin: nothing to create; a non-NULL token so that callers can tell success */
    static int token;
    if (!side) return fail(BPP_E_BADARG, "bpp_side_create: NULL pointer");
    *side = &token;
    return 0;
}

int bpp_side_destroy(void *side) {
    (void)side;
    return 0;
}

int bpp_rollout_uniform_stream(const bpp_batch *b, const bpp_step_out *out, int64_t *actions, uint64_t seed, uint64_t step0,
                               int32_t nsteps, const bpp_stream *s, int32_t refill_every, void *side, void *stream) {
    (void)side;
    if (!b || !s) return fail(BPP_E_BADARG, "bpp_rollout_uniform_stream: NULL pointer");
    if (b->pool_mode != BPP_POOL_RING || refill_every < 1 || refill_every > s->depth - 3)
        return fail(BPP_E_BADARG, "bpp_rollout_uniform_stream: needs a ring pool and 1 <= refill_every <= depth - 3");
    int rc = 0;
    for (int32_t done = 0; rc == 0 && done < nsteps; done += refill_every) {
        int32_t n = nsteps - done < refill_every ? nsteps - done : refill_every;
        rc = bpp_rollout_uniform(b, out, actions, seed, step0 + (uint64_t)done, n, stream);
        if (rc == 0) rc = bpp_stream_refill(s, stream);
    }
    return rc;
}
