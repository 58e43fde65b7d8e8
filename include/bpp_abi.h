/*
 * bpp_abi.h -- C ABI of the MI355X-native vectorised 3D bin-packing environment step.
 *
 * The reference (alexfrom0815/Online-3D-BPP-DRL) is pure Python and has no FFI; this header is the
 * boundary a maintainer would bind with ctypes.  Every entry point names the
 * reference code it replaces (paths relative to the reference root).
 *
 * Two shared libraries export exactly these symbols:
 *   libbpp_hip.so     (online-3d-bpp-drl_amd/csrc)  all pointers are DEVICE pointers, `stream` is a
 *                                                   hipStream_t; the product.
 *   libbpp_oracle.so  (oracle/)                     all pointers are HOST pointers, `stream` ignored;
 *                                                   test infrastructure only (the parity checker).
 *
 * Ownership: the caller allocates and owns every buffer (PyTorch-ROCm tensors in practice); the
 * library never allocates or frees device memory.  Process-global state: the thread-local last-error
 * string and the launch-shape knobs (bpp_knobs, which never change results); nothing else -- the side stream
 * and events of bpp_rollout_uniform_stream's overlapped schedule belong to the caller (bpp_side_create /
 * bpp_side_destroy, ABI v14; rounds 3-4 kept one set per device inside the library).  Kernels are
 * enqueued on `stream` and the calls return without synchronising.
 *
 * Error convention: 0 = success; >0 = hipError_t from the runtime; <0 = BPP_E_* below.  A message
 * is available from bpp_last_error().  An infeasible or out-of-range *action* is not an error: it
 * ends the episode with reward 0 exactly as envs/bpp0/bin3D.py:108-112 does.
 */
#ifndef BPP_ABI_H
#define BPP_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPP_ABI_VERSION 16

#define BPP_E_BADARG   (-1)  /* NULL pointer, non-positive size, unknown rule ... */
#define BPP_E_TOOLARGE (-2)  /* W*L or H beyond what the kernels support (see bpp_limits) */

/* Which feasibility rule a mask is built with. */
#define BPP_RULE_UTILS 0 /* acktr/utils.py:8-35   check_box -- the mask the ACKTR loop consumes   */
#define BPP_RULE_SPACE 1 /* envs/bpp0/space.py:111-144 Space.check_box -- the placement rule, also
                            used by PackingGame.get_possible_position (envs/bpp0/bin3D.py:72-93)  */

/* bpp_step action meaning "leave this bin alone" (no reference counterpart: the reference steps its envs one at a
 * time; lookahead searches step a SUBSET of a batch): state, heightmap and Monitor sums stay as
 * they are, reward 0, done 0, and the bin's current observation and mask are written again. */
#define BPP_ACTION_NOOP INT64_MIN

/* bpp_reset modes */
#define BPP_RESET_INIT    0 /* first reset: episode index 0                                        */
#define BPP_RESET_ADVANCE 1 /* later VecEnv.reset(): every bin abandons its episode, next sequence */

/* Per-bin scalar state, 48 bytes, array-of-structs so one bin's record is three 16-byte accesses.
 * Replaces the Python objects hanging off one PackingGame + its bench.Monitor wrapper.  The last four
 * words cache pool entries (packed x | y<<8 | z<<16) so that a step needs no dependent pool lookup. */
typedef struct bpp_env_state {
    int32_t cursor;   /* index of `next_box` in the current sequence: BoxCreator FIFO head,
                         envs/bpp0/binCreator.py:15-22, envs/bpp0/bin3D.py:68-70                   */
    int32_t episode;  /* episodes finished by this bin (= creator resets - 1)                      */
    int32_t n_boxes;  /* len(space.boxes), envs/bpp0/space.py:23,176                              */
    int32_t vol_sum;  /* sum of x*y*z over placed boxes, envs/bpp0/space.py:146-151                */
    double  ep_ret;   /* bench.Monitor running sum(self.rewards), baselines/bench/monitor.py:58-62 */
    int32_t ep_len;   /* bench.Monitor len(self.rewards), baselines/bench/monitor.py:63            */
    int32_t seq;      /* pool row this episode plays = (global bin id + episode*env_id_total) mod P */
    uint32_t item_cur;   /* pool[seq][min(cursor, T-1)]: BoxCreator.preview(1)[0], binCreator.py:15-18  */
    uint32_t item_next;  /* pool[seq][min(cursor+1, T-1)]: the item shown after a successful placement  */
    uint32_t item_reset; /* pool[next episode's row][0]: the item shown after a failed placement        */
    uint32_t hmax;       /* height of the bin's highest cell, max(space.plain): lets the 20x20 kernel pick the
                            one-word histogram for bins that are still low (csrc/bpp_tile_kernel.inl: kLowTop)   */
} bpp_env_state;

/* One shard of bins living on one device.  Replaces N x (PackingGame + Space + BoxCreator +
 * Monitor) behind ShmemVecEnv (acktr/envs.py:77-118, baselines/common/vec_env/shmem_vec_env.py). */
typedef struct bpp_batch {
    int32_t num_envs;      /* E: bins in this shard                                               */
    int32_t W, L, H;       /* container_size, envs/bpp0/bin3D.py:10-16                             */
    int32_t rotation;      /* enable_rotation: action/mask length M = W*L*(1+rotation), :35-38     */
    int32_t mask_rule;     /* rule of the mask written by bpp_reset/bpp_step (BPP_RULE_*)          */
    int32_t pool_size;     /* P: sequences in the pool                                             */
    int32_t pool_len;      /* T: entries per sequence, padded; the LAST entry of every sequence is
                              a terminator item and is what a cursor >= T keeps returning          */
    int64_t env_id_base;   /* global id of this shard's bin 0 (multi-GPU: rank * E)                */
    int64_t env_id_total;  /* bins in the whole job; episode k of global bin g plays sequence
                              (g + k * env_id_total) mod P, independent of the GPU count           */
    const uint8_t *seq_pool; /* [P][T][4] = (x, y, z, 0) item sizes                                */
    uint8_t *hmap;         /* [E][W*L] Space.plain held as BYTES (heights <= H <= 255; the reference's
                              int32 only exists at the contract edge: obs plane 0 carries the same
                              values as float32), row-major idx = lx*L + ly,
                              envs/bpp0/space.py:22,153-156                                        */
    bpp_env_state *state;  /* [E]                                                                  */
    double *ep_acc;        /* NULL, or [E][4] per-bin episode accumulators kept by bpp_step: when bin e
                              finishes an episode its row gets += (return, final ratio, length, 1) --
                              a plain read-modify-write by the one lane that owns the bin, no atomics,
                              so every row is the float64 sum of that bin's episodes in the order it
                              played them.  bpp_episode_acc_reduce sums the rows in a fixed order: the
                              four job-level sums (main.py:159-162) are bit-reproducible             */
    int32_t pool_mode;     /* BPP_POOL_STATIC: the row rule above.  BPP_POOL_RING: seq_pool is the ring of a
                              bpp_stream, pool_size = depth * num_envs, episode k of LOCAL bin e plays row
                              (k mod depth) * num_envs + e, refilled by bpp_stream_refill.  A ring row starts
                              with TWO LOOK-AHEAD ENTRIES -- item 1 of the bin's NEXT row and item 0 of the row
                              after that, copied there when those rows are cut -- and holds item i at entry
                              2 + i (pool_len counts all entries): what a step fetches ahead for both outcomes
                              (the item two places on; the first items of the next two episodes,
                              binCreator.py:15-18) then comes from the ONE row it is reading anyway instead
                              of three rows scattered over a ring far larger than the caches               */
    int32_t reserved0;
    void *seq_cache;       /* NULL, or (BPP_POOL_RING only, depth >= 5, pool_len < 8192) BPP_SEQ_CACHE_BYTES(num_envs) bytes,
                              128-byte aligned, contents opaque, ZERO-FILLED by the caller when it is allocated and again
                              whenever `state` or the ring are written behind the library's back (restoring a
                              checkpoint, cloning bins): the ROW CACHE of a ring pool.  A read that misses every cache
                              takes longer than a step workgroup lives once the step kernel's write stream saturates
                              the memory, so one lane looking ahead into a ring far larger than the caches holds its
                              whole workgroup back.  With a row cache the 10x10 / 20x20 step kernels keep, per bin, two
                              128-byte lines with what its next steps look ahead to; a bin that moves to another row
                              posts a request, copier workgroups at the front of the NEXT step launch read the ring
                              for it, and the step after that finds the new line.  Whatever a line cannot answer is read
                              from the ring as before: results are the same with and without the cache (every other
                              kernel simply drops the bin's lines).  A line refers to the row after next and is built a
                              step after it was asked for: refill at least every depth - 4 lock-steps (depth - 3
                              without a cache).                                                                        */
} bpp_batch;

#define BPP_SEQ_CACHE_BYTES(E) ((size_t)(E) * (2 * 128 + 8 + 8))

#define BPP_POOL_STATIC 0
#define BPP_POOL_RING   1

/* Endless CUT-2 item supply generated on the device (removes the finite pool).  The reference
 * draws one sequence after the other from the worker's `random` stream (envs/bpp0/mdCreator.py:147-166).  Here
 * every bin owns an exact random.Random(seed0 + global bin id) (MT19937 state in device memory) and episode k of
 * the bin plays the k-th sequence that stream yields through the reference creator.  `ring` holds `depth` rows per
 * bin; bpp_stream_refill cuts new sequences into the rows of episodes the bin has finished, so that `depth`
 * episodes are available from the current one.  A step reads rows up to two episodes ahead and a bin can finish at
 * most one episode per step: refill at least every depth - 3 lock-steps.  Rows are two look-ahead entries (see
 * bpp_batch.pool_mode) followed by the items, padded with the terminator (W,L,H); a sequence longer than pool_len - 3 is
 * truncated and counted in `overflow` (size pool_len as
 * W*L*H / bound_lo^3 + 3 to make that impossible -- with rows that long, at most 2048 entries, the refill is the
 * four-kernel pipeline scan / pretwist / cut / sort of csrc/bpp_stream_gen.inl, else one lane per bin).
 * `mt` and `work` are opaque; bpp_stream_sizes tells how large they must be (both 16-byte aligned).  `mt` is an
 * array of num_envs equal records, one per bin (copy a bin's record together with its ring rows and gen_next to
 * clone its item stream; checkpoint the whole buffer); `work` is scratch between calls.
 *
 * rng = BPP_STREAM_RNG_COUNTER: the same cutting algorithm (mdCreator.py:59-138, statement for statement) drawing from
 * a COUNTER-BASED generator instead of CPython's Mersenne Twister -- the survey's bar for this row is distribution
 * parity ("RNG streams can't match Python's random"); the exact stream above exceeds it and pays for it: a 6.6 KB state
 * per bin, a kernel that regenerates states, rejection loops over a serial output stream.  Here a sequence is a pure
 * function of (seed0, the bin's stream id, the episode index): no state, no regeneration, every draw independent of the
 * one before.  Normative definition (the oracle library, the device kernels and sequences.CounterRandom follow it):
 *     fmix32(x):  x ^= x >> 16;  x *= 0x85ebca6b;  x ^= x >> 13;  x *= 0xc2b2ae35;  x ^= x >> 16        (uint32)
 *     key of episode k of stream id sid (uint64, initially the bin's global id):
 *         h = fmix32((uint32)seed0 + 0x9E3779B9);  h = fmix32(h ^ (uint32)(seed0 >> 32));
 *         h = fmix32(h ^ (uint32)sid);             h = fmix32(h ^ (uint32)(sid >> 32));
 *         klo = fmix32(h ^ k);                     khi = fmix32(klo + 0x7F4A7C15 + k)
 *     word(n, a) = fmix32((klo + n * 0x9E3779B9) ^ (khi + a * 0x85EBCA77))            n = index of the draw, a = attempt
 *     below(lim), the n-th draw of the sequence (random.choice(flags) = flags[below(len)], random.randint(1, v) = 1 +
 *     below(v)):  a = 0;  m = (uint64)word(n, a) * lim;  if (uint32)m < lim:  t = (2^32 - lim) mod lim;  while (uint32)m <
 *     t:  a += 1, m = (uint64)word(n, a) * lim;   result = m >> 32          (Lemire's unbiased multiply-shift)
 * (The per-stream key passes through ONE 32-bit hash h and every word is one fmix32 permutation of a 32-bit input: keys
 * are distinct for sid < 2^32 under one seed, but all bins draw from the same 2^32-point function at structured offsets --
 * the quality of the streams rests on fmix32's avalanche alone.  Accepted for a benchmark / training item supply; checked
 * by the distribution test and by tests/test_stream_counter.py's correlation test across neighbouring bins, episodes and
 * draws.  Not a cryptographic or a 2^64-period generator.)
 * A bin's record is then 16 bytes: its stream id (copied with the record when a bin is cloned, so that the copy
 * continues the SOURCE's stream).  Uniformity of every draw is what makes the distribution of sequences the
 * reference's; tests/test_stream_counter.py compares item-size and sequence-length statistics of the two generators. */
#define BPP_STREAM_RNG_MT19937 0
#define BPP_STREAM_RNG_COUNTER 1
typedef struct bpp_stream {
    int32_t num_envs;      /* E                                                                    */
    int32_t depth;         /* D >= 4: ring rows per bin                                           */
    int32_t pool_len;      /* T: entries per row                                                  */
    int32_t W, L, H;
    int32_t bound_lo, bound_hi;   /* MDlayerBoxCreator(container_size, [bound_lo, bound_hi])      */
    int64_t env_id_base;   /* global id of local bin 0                                            */
    uint64_t seed0;
    uint8_t *ring;         /* [D][E][T][4] == bpp_batch.seq_pool                                   */
    uint32_t *mt;          /* [E][sizes[0] / E] generator records (opaque)                         */
    void *work;            /* sizes[1] bytes of scratch (opaque)                                   */
    int32_t *gen_next;     /* [E] next episode index to be generated                               */
    const bpp_env_state *state; /* [E] == bpp_batch.state (read: episode)                          */
    int32_t *overflow;     /* NULL or [1]: incremented per truncated sequence                      */
    int32_t rng;           /* BPP_STREAM_RNG_MT19937 (exact CPython stream) / BPP_STREAM_RNG_COUNTER            */
    int32_t reserved1;
} bpp_stream;

/* out[0] = uint32 words of `mt`, out[1] = bytes of `work` for the geometry in *s (num_envs, depth, pool_len, W, L, H
 * and the bounds filled in; the pointers are not looked at). */
int bpp_stream_sizes(const bpp_stream *s, int64_t out[2]);
/* Seed every bin's generator; gen_next = 0.  Call once, then bpp_stream_refill, then bpp_reset. */
int bpp_stream_init(const bpp_stream *s, void *stream);
/* Generate until gen_next[e] == state[e].episode + depth for every bin (state must be initialised or zero). */
int bpp_stream_refill(const bpp_stream *s, void *stream);

/* Outputs of one lock-step.  Layout = what VecPyTorch hands the ACKTR loop (acktr/envs.py:170-193)
 * plus the location mask the loop builds per observation (main.py:122-129,163-169). */
typedef struct bpp_step_out {
    float   *obs;      /* [E][4*W*L] float32: planes hmap, x, y, z; envs/bpp0/bin3D.py:49-66; after a
                          terminal step it is the NEXT episode's first observation
                          (baselines/common/vec_env/shmem_vec_env.py:126-130)                      */
    float   *mask;     /* [E][M] float32 0/1 for `obs`; all-ones when nothing is feasible
                          (acktr/utils.py:59-60,91-92).  May be NULL (mask not wanted).            */
    float   *reward;   /* [E] float32( float64(x*y*z / (W*L*H)) * 10 ), 0 on failure; bin3D.py:108-121 */
    uint8_t *done;     /* [E] 1 iff the placement failed (episode over); bin3D.py:108-112          */
    int32_t *counter;  /* [E] info['counter'] of the bin the action was applied to; bin3D.py:111,124 */
    double  *ratio;    /* [E] info['ratio'] (before any auto-reset); bin3D.py:111,125              */
    double  *ep_ret;   /* [E] where done: info['episode']['r'] before round(.,6); else running sum */
    int32_t *ep_len;   /* [E] where done: info['episode']['l']; else running length                */
    int64_t *next_action; /* NULL, or [E]: bpp_step additionally draws, from the mask it just produced, the
                          action bpp_sample_feasible(mask, ., sample_seed, sample_step) would return --
                          the uniform-feasible policy fused into the step (no reference counterpart;
                          benchmark/soak driver).  May alias the `actions` argument.               */
    uint64_t sample_seed;
    uint64_t sample_step;
    float   *host_reward; /* NULL, or [E] in page-locked HOST memory mapped into the device (hipHostMalloc; PyTorch's pinned   */
    uint8_t *host_done;   /* memory): bpp_step ALSO writes reward / done there, so that the host side of VecEnv.step_wait()   */
                          /* (acktr/envs.py:189-193: CPU reward tensor, numpy done) needs no copy, only bpp_wait(stream).     */
                          /* Both or neither.                                                                                  */
} bpp_step_out;

/* Launch-shape tuning knobs of the step/reset/mask kernels (no reference counterpart).  Process-global;
 * initialised once from the environment (BPP_EPW, BPP_WPB, BPP_XCD, BPP_FORCE_GENERIC, BPP_ABLATE) the first
 * time they are needed, afterwards only bpp_set_knobs changes them -- a launch never reads the environment.
 * Results never depend on them (tests/test_gpu_parity.py replays the golden rollouts under every setting). */
typedef struct bpp_knobs {
    int32_t bins_per_wave;    /* 0 = heuristic (10x10: 4, 20x20: 1); else 1..64 (rounded down to a power of two) */
    int32_t waves_per_group;  /* 0 = default (4); else 1..16                                                    */
    int32_t xcd_remap;        /* 1 = give every XCD one contiguous eighth of the bins (default), 0 = off        */
    int32_t force_generic;    /* 1 = route every geometry through the generic cell-scan kernel                  */
    int32_t ablate;           /* profiling builds only (-DBPP_ENABLE_ABLATION): phase bit mask, else ignored    */
    int32_t legacy_fast;      /* 1 = run the runtime-geometry prefix-image kernel (bpp_fast_kernel) also for the
                                 10x10 / 20x20 bins that have a compiled tile kernel (bpp_tile_kernel)          */
    int32_t tile_groups;      /* bpp_tile_kernel: groups of bins a wave walks through, 1 / 2 / 4; 0 = by size (1, or 2 / 4
                                 once a launch's outputs exceed the Infinity Cache) */
    int32_t stream_legacy;    /* bpp_stream_refill: 1 = the one-lane-per-bin refill kernel also where the four-kernel
                                 pipeline (scan / pretwist / cut / sort) applies; counter generator only: 2 = scan / cut
                                 per bin / sort where the rows pipeline (scan / cut and rank per sequence) applies,
                                 3 = the rows pipeline with list and staging capacities so small that most rows take
                                 its second attempt (tests)                                                               */
    int32_t stream_overlap;   /* bpp_rollout_uniform_stream: 1 (default) = with depth >= 2 * refill_every + 3 the refills
                                 run on a high-priority side stream beside the next refill_every lock-steps, 0 = on
                                 the caller's stream between the lock-steps                                    */
    int32_t reserved[3];
} bpp_knobs;
int bpp_get_knobs(bpp_knobs *out);
int bpp_set_knobs(const bpp_knobs *k);

/* Which kernel and launch shape bpp_step / bpp_reset / bpp_mask_* use for a geometry under the current knobs
 * (documentation and tests): out = {kernel, histogram words K, bins per wave, waves per workgroup, workgroups,
 * LDS bytes per workgroup}. */
#define BPP_KERNEL_CELLSCAN  0 /* bpp_kernel: any W*L <= 1024, per-candidate window scan                  */
#define BPP_KERNEL_PREFIX_RT 1 /* bpp_fast_kernel: prefix image, runtime geometry (W*L % 4 == 0, H <= 22) */
#define BPP_KERNEL_TILE      2 /* bpp_tile_kernel: prefix image, compile-time 10x10 / 20x20 geometry      */
int bpp_launch_info(int32_t E, int32_t W, int32_t L, int32_t H, int32_t rotation, int32_t out[6]);

int bpp_abi_version(void);
const char *bpp_last_error(void);
/* out[0] = max W*L supported, out[1] = max H supported. */
int bpp_limits(int32_t out[2]);

/* VecEnv.reset(): zero every heightmap, restart every creator/Monitor, emit first obs (+mask).
 * Replaces PackingGame.reset (envs/bpp0/bin3D.py:55-59) over all workers
 * (baselines/common/vec_env/shmem_vec_env.py:61-67).  Only obs and mask of `out` are written. */
int bpp_reset(const bpp_batch *b, int32_t mode, const bpp_step_out *out, void *stream);

/* VecEnv.step(actions): one lock-step of all bins, fused: action decode incl. the `idx > area`
 * rotation quirk (bin3D.py:96-105), placement rule (space.py:111-144), heightmap update
 * (space.py:36-46,164-181), reward (bin3D.py:44-46,114-121), info (bin3D.py:111,123-125), Monitor
 * (baselines/bench/monitor.py:51-77), auto-reset (shmem_vec_env.py:126-130), next observation
 * (bin3D.py:61-66) and its feasibility mask (acktr/utils.py:37-94); finished episodes are added to the bins' rows of
 * bpp_batch.ep_acc (main.py:159-162); with out->host_reward / host_done set, reward and done are ALSO written into that
 * mapped host memory (acktr/envs.py:189-193 hands the loop a CPU reward tensor and a numpy done).  actions: [E] int64. */
int bpp_step(const bpp_batch *b, const int64_t *actions, const bpp_step_out *out, void *stream);


/* Batched drop-in for acktr.utils.get_possible_position (rotation=0, acktr/utils.py:37-62) and
 * get_rotation_mask (rotation=1, :64-94) on a [E][4*W*L] float32 observation batch. */
int bpp_mask_from_obs(const float *obs, float *mask, int32_t E, int32_t W, int32_t L, int32_t H,
                      int32_t rotation, int32_t rule, void *stream);

/* Same from a raw int32 heightmap batch + items [E][3] (x,y,z): rule=BPP_RULE_SPACE, rotation=0
 * is PackingGame.get_possible_position (envs/bpp0/bin3D.py:72-93). */
int bpp_mask_from_hmap(const int32_t *hmap, const int32_t *items, float *mask, int32_t E, int32_t W,
                       int32_t L, int32_t H, int32_t rotation, int32_t rule, void *stream);

/* Benchmark/test action source (no reference counterpart; SURVEY.md 8d "actions for timing"):
 * uniform choice among mask==1 entries with a counter-based RNG keyed by (seed, global bin id,
 * step), a 32-bit multiply-xorshift hash: pick = hash * count >> 32, the pick-th set entry in index
 * order.  actions: [E] int64. */
int bpp_sample_feasible(const float *mask, int64_t *actions, int32_t E, int32_t M, int64_t env_id_base,
                        uint64_t seed, uint64_t step, void *stream);

/* Failure-path variant of that action source (SURVEY.md 8d "variant: epsilon = 1 % uniformly random over all actions to
 * exercise the failure path"; no reference counterpart): with probability eps_q24 / 2^24 the action of bin e is REPLACED by a
 * uniform draw over all M entries, feasible or not.  Normative: with hash32 the function of bpp_sample_feasible,
 *     coin = hash32(seed ^ BPP_EPS_KEY_COIN, global bin id, step) >> 8;   if coin < eps_q24:
 *     actions[e] = (hash32(seed ^ BPP_EPS_KEY_PICK, global bin id, step) * M) >> 32
 * so a (seed, step) pair used for bpp_sample_feasible can be reused here.  eps_q24 in 0 .. 2^24; 0 enqueues nothing. */
#define BPP_EPS_KEY_COIN 0x5851F42D4C957F2DULL
#define BPP_EPS_KEY_PICK 0xDA942042E4DD58B5ULL
int bpp_epsilon_override(int64_t *actions, int32_t E, int32_t M, int64_t env_id_base, uint64_t seed, uint64_t step,
                         uint32_t eps_q24, void *stream);

/* Masked categorical action selection of the policy head, fused.  Replaces the
 * inference half of acktr.distributions.Categorical.forward (acktr/distributions.py:71-84) as used by
 * Policy.act (acktr/model.py:56-68) after the linear layer:
 *     lx = softmax(logits - 14 * (1 - mask)) + 1e-5 ;  probs = lx / sum(lx)      (FixedCategorical(probs=lx))
 *     action  = argmax(probs)                      if deterministic            (dist.mode())
 *             = inverse-CDF draw with u in [0,1)   otherwise                   (dist.sample(); the
 *               reference draws with torch.multinomial, whose RNG stream is not reproducible -- here
 *               u = (hash32(seed, global bin id, step) >> 8) * 2^-24, a counter-based stream)
 *     log_prob = log(clamp(probs[action], eps, 1 - eps)), eps = 2^-23           (dist.log_probs(action))
 * float32 arithmetic; logits, mask: [E][M] float32; action: [E] int64; log_prob: [E] float32 (may be NULL). */
int bpp_masked_act(const float *logits, const float *mask, int64_t *action, float *log_prob, int32_t E, int32_t M,
                   int64_t env_id_base, uint64_t seed, uint64_t step, int32_t deterministic, void *stream);
/* The same selection with (seed, step) read from DEVICE memory -- seed_step: uint64 [2] -- when the kernel runs (v16).  For loops
 * captured in a HIP graph: a captured launch replays its arguments, so a `step` passed by value would give every replay the same
 * draws; here the caller advances seed_step[1] on the device between two replays (one more node of the same graph, e.g. a torch
 * `add_`) and every replay draws afresh.  At the reference's own scale (16 ... 1 024 bins) a policy step is launch-bound -- ~25
 * launches of a few microseconds each -- and one graph launch replaces them (examples/rollout_with_policy.py --graph). */
int bpp_masked_act_counter(const float *logits, const float *mask, int64_t *action, float *log_prob, int32_t E, int32_t M,
                           int64_t env_id_base, const uint64_t *seed_step, int32_t deterministic, void *stream);

/* Training half of the same head (acktr/distributions.py:71-101 as consumed by Policy.evaluate_actions,
 * acktr/model.py:90-96), forward and backward, float32:
 *     log_prob[e] = dist.log_probs(action)[e]            (log of the clamped normalised probability of action[e])
 *     entropy[e]  = dist.entropy()[e]                    (the loop takes its mean)
 *     bad_prob[e] = sum_k softmax(logits)[e,k] * (1 - mask[e,k])   (row sums of `bx`; `prob_loss` is their sum / (E*M))
 * bpp_masked_evaluate_backward writes d(sum_e g_log_prob[e]*log_prob[e] + g_entropy[e]*entropy[e] + g_bad_prob[e]*bad_prob[e])
 * / d logits into grad_logits [E][M].  One wave per bin, any M. */
int bpp_masked_evaluate(const float *logits, const float *mask, const int64_t *action, float *log_prob, float *entropy,
                        float *bad_prob, int32_t E, int32_t M, void *stream);
int bpp_masked_evaluate_backward(const float *logits, const float *mask, const int64_t *action, const float *g_log_prob,
                                 const float *g_entropy, const float *g_bad_prob, float *grad_logits, int32_t E, int32_t M,
                                 void *stream);

/* Host-side CUT-2 item-sequence generator, multithreaded.  Restates
 * envs/bpp0/mdCreator.py:59-166 (Box.benchmark_split, bin.gen_benchmark incl. its iterate-while-mutating
 * list walk, depart_box) and draws from an exact re-implementation of CPython's `random.Random(seed)`
 * (MT19937, init_by_array, getrandbits-based randbelow), so sequence k equals what the reference's
 * MDlayerBoxCreator produces under random.seed(seed0 + k).  pool: HOST buffer [n][T][4] uint8, rows padded
 * with the terminator (W,L,H).  Returns 0, or BPP_E_TOOLARGE when a sequence does not fit in T-1 entries
 * (lengths[k] always receives the true length).  Both libraries export it; no GPU is involved. */
int bpp_gen_cut2(uint8_t *pool, int32_t *lengths, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H,
                 int32_t bound_lo, int32_t bound_hi, uint64_t seed0, int32_t threads);

/* CUT-1: restates envs/bpp0/cutCreator.py:32-128 (CuttingBoxCreator: guillotine cuts until
 * every piece lies inside box_range = {low_x, low_y, low_z, high_x, high_y, high_z}, then repeated random draws
 * among the pieces whose support is complete).  Sequence k consumes exact re-implementations of
 * random.Random(seed0 + k) and -- for the rotation coin, np.random.rand() >= 0.5 swaps x and y, :119-125 --
 * numpy's legacy RandomState(seed0 + k), so it equals what the reference creator yields after
 * random.seed(s); np.random.seed(s); reset().  Same pool layout and return codes as bpp_gen_cut2; BPP_E_BADARG
 * also when the reference's own `assert pos_range[0] <= pos_range[1]` would fire.  Seeds must stay below 2^32. */
int bpp_gen_cut1(uint8_t *pool, int32_t *lengths, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H,
                 const int32_t box_range[6], int32_t rotation, uint64_t seed0, int32_t threads);

/* RS: restates envs/bpp0/binCreator.py:24-40 (RandomBoxCreator): row k = the first T-1 draws
 * box_set[np.random.randint(0, n_box)] of numpy's legacy RandomState(seed0 + k), then the terminator (W,L,H).
 * box_set: HOST int32 [n_box][3]. */
int bpp_gen_rs(uint8_t *pool, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H, const int32_t *box_set, int32_t n_box,
               uint64_t seed0, int32_t threads);

/* VecEnv.step_wait()'s host side (acktr/envs.py:189-193 hands the loop a CPU reward tensor and a numpy `done`): copy
 * `nbytes` from device memory to (page-locked) host memory behind everything enqueued on `stream` and wait for it --
 * hipMemcpyAsync + hipStreamSynchronize in one call.  BppVecEnv.step() fetches reward + done (5 bytes per bin) with
 * it.  The only entry point that blocks the calling thread. */
int bpp_fetch_to_host(const void *device_src, void *host_dst, int64_t nbytes, void *stream);
/* hipStreamSynchronize(stream): what step_wait() needs when bpp_step wrote reward / done into host memory itself
 * (bpp_step_out.host_reward / host_done). */
int bpp_wait(void *stream);
/* The same wait without the runtime's synchronisation call (v15): bpp_mark enqueues "store `value` into the 32-bit word at
 * `host_flag` (page-locked host memory the device can write) once everything enqueued on `stream` before is complete" -- a
 * stream memory operation, or a one-thread kernel where the runtime offers none --, bpp_wait_mark spins on that word (and asks
 * the stream for errors now and then; a stream that reports completion without the word having arrived is synchronised and the
 * word read once more).  BppVecEnv.step_async() marks its staging buffer with the step's serial number behind the step kernel
 * (and the eager gather), step_wait() waits for it: the host sees reward / done a few microseconds after the kernel's last
 * store instead of after a signal round trip.  Use a value the word does not hold yet. */
int bpp_mark(void *host_flag, uint32_t value, void *stream);
int bpp_wait_mark(const void *host_flag, uint32_t value, void *stream);
/* Visibility: the word and the data it announces (reward / done / the eager gather's arrays) are plain stores into page-locked
 * host memory; the host may read the data as soon as it sees the word only if that memory is host-COHERENT (hipHostMalloc's
 * default; torch's pin_memory()).  bpp_mark asks the runtime for the flag's allocation flags once per buffer address range; for
 * memory allocated hipHostMallocNonCoherent it enqueues the marker KERNEL instead of the stream memory operation (its store is a
 * system-scope release behind a system-scope fence), and bpp_wait_mark ends with a stream synchronisation for such a flag. */

/* `infos` of the bins that finished in a lock-step (main.py:159-162 reads infos[i]['episode']['r'] and infos[i]['ratio']
 * of exactly those; bench/monitor.py:64-75, bin3D.py:111).  `n` = the number of finished bins the caller counted in ITS
 * copy of `done` (step_wait() has it on the host already).  One launch compacts the step's per-bin outputs of the bins
 * with done != 0, in ascending bin order, into `dev` (device memory, BPP_FINISHED_BYTES(E) bytes) as five arrays of n
 * entries behind a 32-byte header:
 *     int32 count (the kernel's own count of finished bins), 28 bytes reserved;
 *     float64 ep_ret[n] (Monitor's `r` before round(., 6)); float64 ratio[n]; int32 ep_len[n] (Monitor's `l`);
 *     int32 counter[n] (boxes placed); int32 bin[n] (local bin numbers);
 * then the first BPP_FINISHED_BYTES(n) bytes are copied to `host` (page-locked) and the stream is synchronised: only what
 * is needed crosses PCIe, in one transfer, already laid out as the arrays a caller wants.  dev == NULL: `host` is
 * page-locked memory MAPPED into the device and the kernel writes header and arrays there itself (no staging copy).  count != n (the caller's `done`
 * belongs to another step): BPP_E_BADARG.  Blocks the calling thread like bpp_fetch_to_host. */
/* n == BPP_GATHER_ENQUEUE_ONLY (dev must be NULL, `host` mapped page-locked memory of BPP_FINISHED_BYTES(E) bytes): the EAGER form
 * -- the compaction is only enqueued behind the step, with the five arrays laid out for E entries (ep_ret at 32, ratio at 32 + 8 E,
 * ep_len at 32 + 16 E, counter at 32 + 20 E, bin at 32 + 24 E); nothing is copied, nothing waited for, nothing checked: the caller's
 * own synchronisation of the step (bpp_wait) covers it, the header's count then says how many entries each array holds.  One more
 * launch per step (~10 us of device time at 65 536 bins) instead of a launch AND a second synchronisation when `infos` is read. */
#define BPP_GATHER_ENQUEUE_ONLY (-1)
#define BPP_FINISHED_BYTES(n) (32 + 28 * (int64_t)(n) + 4)      /* (+ 4: room for the 8-byte alignment of nothing -- n may be odd) */
int bpp_gather_finished(const uint8_t *done, const double *ep_ret, const double *ratio, const int32_t *ep_len,
                        const int32_t *counter, int32_t E, void *dev, void *host, int32_t n, void *stream);

/* VecEnv.step_async() of the reference-shaped path in ONE host call (v16; replaces baselines/common/vec_env/shmem_vec_env.py:69-79's
 * one pipe message per worker, acktr/envs.py:182-188): bpp_step(b, actions, out); then, fin_host != NULL, the eager
 * bpp_gather_finished of THAT step's outputs (out->done / ep_ret / ratio / ep_len / counter, BPP_GATHER_ENQUEUE_ONLY) into
 * fin_host; then, host_flag != NULL, bpp_mark(host_flag, value).  Same results as the three calls one after the other; what it
 * saves is two trips through the caller's FFI per lock-step, which is what a step costs at the reference's own scale (16 ... 1024
 * bins: the kernels take a few microseconds).  Nothing is waited for. */
int bpp_step_dropin(const bpp_batch *b, const int64_t *actions, const bpp_step_out *out, void *fin_host, void *host_flag,
                    uint32_t value, void *stream);

/* Policy-free lock-step driver for benchmarks and soak tests (no reference counterpart): enqueues
 * `nsteps` iterations of { bpp_sample_feasible(out->mask -> actions, step0 + t); bpp_step(actions -> out) }
 * on `stream` from one host call.  out->mask must hold the mask of the current observations (as left
 * by bpp_reset / bpp_step) and is required; actions: [E] int64 scratch that ends up holding the last
 * actions taken. */
int bpp_rollout_uniform(const bpp_batch *b, const bpp_step_out *out, int64_t *actions, uint64_t seed,
                        uint64_t step0, int32_t nsteps, void *stream);

/* The same driver writing the outputs of lock-step t into outs[t mod nsets] (nsets >= 1 complete output sets; with
 * nsets * bytes-per-set beyond the 256 MiB Infinity Cache every output byte really goes to HBM -- bench.py's
 * `past_l3` leg) and ALWAYS ending with the fused draw, so that on return `actions` holds the draw for lock-step
 * step0 + nsteps.  flags & BPP_ROLLOUT_CONTINUE: `actions` already holds the draw for step0 (left by the previous
 * call) -- the call then enqueues exactly nsteps launches of the step kernel and nothing else; without it the first
 * action is drawn from `first_mask`, the mask of the current observations (as left by bpp_reset / bpp_step). */
#define BPP_ROLLOUT_CONTINUE 1
/* flags bits 8..31: epsilon of SURVEY.md 8d's failure-path variant as a 24-bit fraction (BPP_ROLLOUT_EPS(0.01 * 2^24) = 1 %):
 * every draw is followed by bpp_epsilon_override with the draw's own (seed, step).  0 = the plain uniform-feasible policy.  The
 * field holds 24 bits: the largest epsilon it carries is (2^24 - 1) / 2^24 -- the macro clamps, so epsilon = 1.0 means "all but one
 * draw in 16.7 million" here, never a wrap-around to 0 (bpp_epsilon_override itself takes the full range 0 .. 2^24). */
#define BPP_ROLLOUT_EPS(q24)      ((int32_t)(((uint32_t)(q24) > 0xffffffu ? 0xffffffu : (uint32_t)(q24)) << 8))
#define BPP_ROLLOUT_EPS_OF(flags) ((uint32_t)(flags) >> 8)
int bpp_rollout_uniform_sets(const bpp_batch *b, const bpp_step_out *outs, int32_t nsets, const float *first_mask,
                             int64_t *actions, uint64_t seed, uint64_t step0, int32_t nsteps, int32_t flags, void *stream);

/* bpp_rollout_uniform over a ring pool: additionally refills the ring after every `refill_every` lock-steps
 * (1 <= refill_every <= depth - 3; depth - 4 with a seq_cache).  With `side` != NULL, depth >= 2 * refill_every + 3 (+ 4) and the
 * stream_overlap knob on, the refills run on the side's high-priority stream BESIDE the following lock-steps (a chunk of
 * lock-steps starts once the refill issued two chunks earlier is complete); everything enqueued on `stream` after the call
 * sees a full ring, as after bpp_stream_refill.  side == NULL: the refills run on `stream` between the lock-steps.  Results are
 * the same either way.
 * bpp_side_create: one high-priority stream + three events on the calling thread's CURRENT device, owned by the caller (one
 * per env; two calls that share a side must be ordered by the caller); bpp_side_destroy(NULL) is a no-op. */
int bpp_side_create(void **side);
int bpp_side_destroy(void *side);
int bpp_rollout_uniform_stream(const bpp_batch *b, const bpp_step_out *out, int64_t *actions, uint64_t seed,
                               uint64_t step0, int32_t nsteps, const bpp_stream *s, int32_t refill_every, void *side, void *stream);

/* Device-side replacement of the training loop's per-bin `infos` scan (main.py:159-162: for every
 * finished episode append info['episode']['r'] and info['ratio'] to the logging deques):
 * acc[0] += sum of episode returns, acc[1] += sum of final ratios, acc[2] += sum of episode lengths,
 * acc[3] += number of episodes, over bins with done != 0.  acc: double[4], caller-zeroed; it is the
 * 32-byte record that multi-GPU jobs all-reduce.
 * Summation order (normative, the oracle library follows it, results are bit-reproducible): BPP_REDUCE_LANES = 1024
 * partial sums, partial r = the contributions of bins r, r + 1024, r + 2048, ... added in ascending order to 0.0 (a
 * bin that is not done contributes nothing); then for d = 512, 256, ..., 1: partial[r] += partial[r + d] for r < d;
 * finally acc[k] += partial[0]. */
#define BPP_REDUCE_LANES 1024
int bpp_episode_stats(const uint8_t *done, const double *ep_ret, const double *ratio, const int32_t *ep_len,
                      int32_t E, double *acc, void *stream);

/* The same four sums from the per-bin accumulators bpp_step keeps (bpp_batch.ep_acc, [E][4]): acc[k] += sum over bins
 * of ep_acc[e][k] in the order stated above (every row contributes); clear != 0 zeroes the rows afterwards.
 * scratch == NULL: one workgroup (all E rows go through one CU's memory pipeline: 61 us for 65 536 bins).
 * scratch != NULL: a caller-owned device buffer of BPP_REDUCE_SCRATCH_BYTES, zero-filled ONCE when it is allocated (its
 * last word is an arrival counter that every call leaves at zero again).  The 1 024 partial sums are then computed by
 * BPP_REDUCE_LANES / 16 workgroups spread over the chip -- each partial still by ONE lane adding its rows in ascending
 * order -- written to the scratch buffer, and the workgroup that arrives last runs the binary tree: the same additions
 * in the same order, bit-identical results, ~10x shorter.  Calls that share a scratch buffer must be ordered (same
 * stream). */
#define BPP_REDUCE_SCRATCH_BYTES (BPP_REDUCE_LANES * 4 * 8 + 64)
int bpp_episode_acc_reduce(double *ep_acc, int32_t E, double *acc, int32_t clear, void *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BPP_ABI_H */
