// bpp_kernels.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI of the vectorised 3D bin-packing
// environment step (include/bpp_abi.h).  Hand-written for wave64; integer/indexing work, no MFMA.
//
// Two implementations of the same step share this file:
//   * bpp_fast_kernel<W,L,K,ROT,MODE>  the production path: compile-time geometry for 10x10 and 20x20 bins,
//                                      W = L = 0 instantiates the same code with runtime geometry for any
//                                      other bin with W*L % 4 == 0 and H <= 22;
//   * bpp_kernel<VEC,MODE>             any W*L <= 1024 (also W*L % 4 != 0, H up to 255), the fallback.
// Work decomposition of the fast path:
//   * a workgroup of 4 waves owns 4*EPW consecutive bins (EPW = 4 for the 10x10 bin, 1 for 20x20); the
//     bins of a wave are contiguous in every tensor ([E][A] byte heightmap, [E][4A] observation, [E][M]
//     mask), so each chunk is streamed with full-width accesses whatever the geometry;
//   * ONE wave per workgroup carries the per-bin scalar chain lane-per-bin (state, action decode,
//     placement rule, reward, Monitor sums, auto-reset, next item; all items come from the state record,
//     no dependent pool lookup) and leaves a small record per bin in LDS; two workgroup barriers are the
//     only synchronisation, workgroups never talk to each other;
//   * every wave then works on its own bins only: placement fill, observation/heightmap store, a
//     packed-histogram prefix image of the tile in LDS, bin-uniform candidate evaluation (item constants
//     in scalar registers, only in-range candidates enumerated), mask store.
//
// Reference semantics are cited next to the code that restates them;
// paths are relative to the reference root.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "../../include/bpp_abi.h"
#include "../../include/bpp_gen.inl"

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#pragma clang fp contract(off)  // float64 reward / return sums must round exactly like numpy

// Where the three pool entries are that the NEXT step may need (binCreator.py:15-18 look-ahead; fetched speculatively for
// both outcomes): item cursor + 2 of the current row, item 1 of the next row, item 0 of the row after that.  A static
// pool: three rows.  The ring of a stream (ring2 == 2): the two entries of the rows behind have been copied into the
// FIRST TWO ENTRIES of the current row when those rows were cut (include/bpp_abi.h), so that all three come from the
// current row -- from ONE line while the cursor is below 28.  (A step's three random HBM reads per bin cost 6 us per
// lock-step of 65 536 bins once pool or ring no longer fit the caches: profiles/r4h_step_kernel_vs_pool_size.json.)
struct LookAheadAt {
    size_t ok, f1, f2;
};
// Profiling aid, compiled in only with -DBPP_ENABLE_ABLATION (tools/build_variant.sh abl -DBPP_ENABLE_ABLATION): BPP_ABLATE=<bit mask>
// then skips individual phases so their cost can be read off rocprofv3 (results are wrong when used).
#ifdef BPP_ENABLE_ABLATION
#define BPP_ABL(p, bit) (((p).ablate & (bit)) != 0)
// Same build: BPP_ABLATE bit 256 makes lane 0 of every wave of every 8th workgroup of bpp_tile_kernel record the
// shader clock at its phase boundaries (read back with bpp_debug_stamps, tools/phase_timeline.py).
constexpr int kStampWaves = 4096;
constexpr int kStampSlots = 16;
__device__ unsigned long long g_stamps[kStampWaves * kStampSlots];
#define BPP_STAMP(p, k)                                                                              \
    do {                                                                                             \
        if (((p).ablate & 256) && lane == 0 && (blockIdx.x & 7u) == 0u && (blockIdx.x >> 3) * 4 + wid < kStampWaves) \
            g_stamps[((blockIdx.x >> 3) * 4 + wid) * kStampSlots + (k)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#else
#define BPP_ABL(p, bit) false
#define BPP_STAMP(p, k) \
    do {                \
    } while (0)
#endif

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kMaxFastWavesPerBlock = 16;  // fast path: wave 0 serves wpb * epw <= 64 bins
constexpr int kMaxArea = 1024;  // W*L
constexpr int kMaxDim = 255;    // W, L, H and item sizes are bytes

enum Mode { kStep = 0, kResetInit = 1, kResetAdvance = 2, kMaskObs = 3, kMaskHmap = 4 };

// n / d for n * d < 2^32 via one v_mul_hi_u32 (m = floor(2^32 / d) + 1); d == 1 has no 32-bit magic
// number (2^32 + 1) and is passed through.
struct FastDiv {
    uint32_t d, m;
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1u ? n : __umulhi(n, m); }
};
FastDiv make_fastdiv(uint32_t d) { return FastDiv{d, d <= 1u ? 0u : (uint32_t)((1ull << 32) / d) + 1u}; }

struct Params {
    // geometry
    int32_t E, W, L, H, A, M, rotation, rule;
    int32_t epw;           // bins per wave
    int32_t lds_per_wave;  // bytes
    int32_t off_mk, off_rec, off_ori, off_P;
    int32_t epw_shift;     // epw == 1 << epw_shift on the fast path
    int32_t xcd_remap;     // 1: XCD-aware block -> bins mapping
    int32_t ablate;        // profiling aid (BPP_ABLATE bit mask): skip a phase to read its cost; results are then wrong
    FastDiv divL, divA, divM, divA4;  // divA4: by A/4 (vector path) or A (scalar path) -> plane index
    FastDiv divW, divM4, divPWW;      // runtime-geometry fast path: by W, M/4 and (L+1)+W
    // sequences
    int32_t P, T, seq_stride, base_mod;  // seq_stride = env_id_total % P, base_mod = env_id_base % P
    int32_t ring2;         // 2 for the ring of a stream (rows start with two look-ahead entries, item i at entry 2 + i), else 0
    double binvol;
    const uint32_t *pool;  // [P][T] packed x | y<<8 | z<<16
    unsigned char *cache;  // bpp_batch.seq_cache (see RowCache) or nullptr
    int32_t ncopy;         // tile step kernel with a row cache: its first ncopy workgroups serve the refresh requests
    // state
    uint8_t *hmap;   // [E][A] bytes
    bpp_env_state *state;
    double *ep_acc;  // [E][4] per-bin episode accumulators or nullptr
    const int64_t *actions;
    // mask-only inputs
    const float *obs_in;
    const int32_t *hmap_in;
    const int32_t *items_in;
    // outputs
    float *obs;
    float *mask;
    float *reward;
    uint8_t *done;
    float *host_reward;   // mirrors of reward / done in mapped host memory, or nullptr
    uint8_t *host_done;
    int32_t *counter;
    double *ratio;
    double *ep_ret;
    int32_t *ep_len;
    // fused uniform-feasible sampling of the next action (bpp_step_out.next_action)
    int64_t *next_action;
    uint64_t sample_seed, sample_step;
    int64_t env_id_base;
};

// Measured on gfx950 this round (profiles/r5e_ubench_sparse_exec_by_instruction.jsonl, tools/ubench sparse): a vector instruction of
// the 4-cycle class (shifts, multiplies, min / max, compares, selects, conversions, three-operand adds, v_readlane ... -- everything
// but v_add / v_sub / v_and / v_xor / v_mov, which issue in ~2.5 cycles) takes ~22 cycles instead of ~4 in a stream of such
// instructions when 8 or fewer of the wave's 64 lanes are active.  Rewriting the kernel's narrow sections (slot words, item read,
// fill offsets, draw decode, statistics sums by all lanes of the bin; candidates spread evenly over the passes) did NOT move
// the kernels (profiles/r5g_*: 28.5 -> 28.6 us, 35.1 -> 35.4 us, 54.1 -> 53.9 us): those sections are short and sit between
// barriers and LDS round trips where the vector pipe is not the limiter.  Kept out of the source; the measurement stays.
// ---- bpp_batch.seq_cache: the row cache of a ring pool ---------------------------------------------------------------
// A read that misses every cache takes 15-18 us under the step kernel's write stream -- longer than a step workgroup
// lives -- so ONE lane waiting for a ring row keeps its workgroup resident past its natural end and the launch pays 6 us
// (profiles/r4s_head_table_experiment, r4x).  With a row cache no step workgroup reads the ring: every bin has two 128-byte
// lines (lines[e][2][32]) holding what its next steps look ahead to, a control word that says which line is current, and a
// request slot.  The bin's deciding lanes post a request when the bin moves to another row (or its cursor nears the end of
// the line's item window); the FIRST ncopy workgroups of the next step launch -- copier workgroups, 1 024 bins each, nothing
// else to do, so their 15 us of waiting holds up nobody -- read the ring rows and write the bin's OTHER line; the step after
// that switches to it.  In between the bin lives on its old line, which also holds the first entries of the next row.
// Anything the current line cannot answer (two rows in two steps, a cache the caller just zeroed, the one episode in
// 65 536 whose tag would be 0) is read from the ring as before: the cache can only make a step faster, never change what it returns.
//   line of (row r of episode k, first item c0), 32 words:
//     [0..7]   entries 0..7 of the bin's NEXT row (its two look-ahead entries, items 0..5)
//     [8],[9]  entries 0, 1 of row r (item 1 of the next row, item 0 of the row after)      [10],[11] unused
//     [12..31] entries 2 + c0 .. 2 + c0 + 19 of row r: items c0 .. c0 + 19 (0 beyond the row)
//   ctl[e] (uint2): x = current line | pending << 1 (0 none, 1 asked for in the previous launch, 2 written meanwhile)
//                       | c0 of line 0 << 3 | c0 of line 1 << 16 (13 bits each)
//                   y = (episode + 1) & 0xffff of line 0 | that of line 1 << 16 (0: no line)
//   req[e] (uint64): 0 = nothing to do; else row | 1 << 31 | (c0 | line << 16) << 32
// Rows: a line is built one step after it is asked for and refers to the row after next: rows up to episode + 3 must exist
// at every step (refill at least every depth - 4 lock-steps).
constexpr int kLineWords = 32, kLineItems = 20, kLineNext = 8;
#ifndef BPP_CACHE_STAT   // (the host emulator of tests/emu counts hits and misses here; nothing in the product)
#define BPP_CACHE_STAT(hit) ((void)0)
#endif
struct RowCache {
    uint32_t *lines;     // [E][2][kLineWords]
    uint2 *ctl;          // [E]
    unsigned long long *req;   // [E]
};
__host__ __device__ __forceinline__ RowCache row_cache(unsigned char *base, int E) {
    RowCache c;
    c.lines = (uint32_t *)base;
    c.ctl = (uint2 *)(base + (size_t)E * 2 * kLineWords * 4);
    c.req = (unsigned long long *)(base + (size_t)E * (2 * kLineWords * 4 + 8));
    return c;
}
// Kernels that do not keep the cache (resets, the runtime-geometry and generic step kernels) drop the bin's lines and any
// request still open: the tile step kernel then reads the ring until its requests have been served again.
__device__ __forceinline__ void row_cache_drop(const Params &p, int e) {
    const RowCache c = row_cache(p.cache, p.E);
    c.ctl[e] = make_uint2(0u, 0u);
    c.req[e] = 0ull;
}
// A copier workgroup of the tile step kernel (its first p.ncopy workgroups) serves the open requests of kCopierBins bins,
// a wave those of 256: it compacts them into a list in its LDS area, then eight lanes build one line (four words each), eight
// requests per iteration, in rounds of 64 whose ring reads are ALL issued before the first line is written -- the reads miss
// every cache, a round costs one such latency (15-18 us), and a round is all a wave ever needs in practice.  Few, fat copier
// workgroups: each holds one of its CU's eight workgroup slots for that long.
// Ordering: a copier may read req[e] in the SAME launch in which the bin's step workgroup posts it (nothing
// orders the two inside a launch) and then builds the line one launch early.  That is benign by construction: (i) the line
// it writes is the bin's OTHER line, which the deciding phase does not read before the launch after next (pending 1 -> 2 ->
// switch), whoever wrote it and when; (ii) what it reads exists: a line for row k + 1 holds entries of rows k + 1 and k + 2,
// whose look-ahead headers were completed when rows k + 3 and k + 4 were cut, and with a row cache the refill schedule
// (every depth - 4 lock-steps) guarantees rows up to episode + 4 at the START of every launch -- the launch in which a bin of
// episode k posts the request included; the regular service one launch later has a row to spare.
constexpr int kCopierBins = 4 * 256;
__device__ __forceinline__ void wave_sync();
__device__ __forceinline__ void row_cache_copier(const Params &p, int e_wave, uint32_t *list) {
    const RowCache c = row_cache(p.cache, p.E);
    const int lane = threadIdx.x & (kWave - 1), T = p.T;
    int n = 0;                                      // requests of this wave's 256 bins: (row | c0 << 32 in two words, bin) triples
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = e_wave + k * kWave + lane;
        const unsigned long long r = e < p.E ? c.req[e] : 0ull;
        const bool valid = ((uint32_t)r >> 31) != 0u;
        const unsigned long long m = __ballot(valid);
        if (valid) {
            const int at = n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            list[3 * at] = (uint32_t)r & 0x7fffffffu;
            list[3 * at + 1] = (uint32_t)(r >> 32);
            list[3 * at + 2] = (uint32_t)e;
        }
        n += __popcll(m);
    }
    wave_sync();
    constexpr int kRound = 8;
    const int slot = lane >> 3, part = lane & 7;    // request within the iteration, four-word part of its line
    for (int base = 0; base < n; base += 8 * kRound) {
        uint32_t v[kRound][4];
#pragma unroll
        for (int i = 0; i < kRound; ++i) {
            const int at = base + 8 * i + slot;
            const bool act = at < n;
            const uint32_t row = act ? list[3 * at] : 0u, c0 = act ? list[3 * at + 1] & 0xffffu : 0u;
            uint32_t rown = row + (uint32_t)p.seq_stride;
            rown = rown >= (uint32_t)p.P ? rown - (uint32_t)p.P : rown;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = part * 4 + j;         // which ring entry goes into word w of the line (see RowCache)
                const uint32_t ent = w < kLineNext ? (uint32_t)w : (w < 10 ? (uint32_t)(w - 8) : 2u + c0 + (uint32_t)(w - 12));
                const bool from_ring = act && w != 10 && w != 11 && ent < (uint32_t)T;
                v[i][j] = from_ring ? p.pool[(size_t)(w < kLineNext ? rown : row) * T + ent] : 0u;
            }
        }
#pragma unroll
        for (int i = 0; i < kRound; ++i) {
            const int at = base + 8 * i + slot;     // (the list again: cheaper than carrying the addresses past the loads)
            if (at < n) {
                const uint32_t buf = (list[3 * at + 1] >> 16) & 1u, e = list[3 * at + 2];
                *(uint4 *)(c.lines + ((size_t)e * 2 + buf) * kLineWords + part * 4) = make_uint4(v[i][0], v[i][1], v[i][2], v[i][3]);
                if (part == 0) c.req[e] = 0ull;     // served
            }
        }
    }
}
__device__ __forceinline__ LookAheadAt look_ahead_at(const Params &p, int seq, int seq_n, int seq_nn, int cursor) {
    const int T = p.T, r2 = p.ring2;
    const bool ring = r2 != 0;
    LookAheadAt a;
    a.ok = (size_t)seq * T + r2 + min(cursor + 2, T - 1 - r2);
    a.f1 = (size_t)(ring ? seq : seq_n) * T + (ring ? 0 : min(1, T - 1));
    a.f2 = (size_t)(ring ? seq : seq_nn) * T + (ring ? 1 : 0);
    return a;
}
// Episode statistics (main.py:159-162): every bin owns one row [return sum, final-ratio sum, length sum, episodes] of
// bpp_batch.ep_acc and the lane that decides the bin adds a finished episode to it with a plain read-modify-write.
// No atomics: a row has exactly one writer per launch, launches on a stream are ordered, so the row is the float64
// sum of the bin's episodes in the order it played them -- deterministic, whatever the launch shape.  (Rounds 1-2
// added into 256 shared slots with float64 L2 atomics; one GPU-suite run lost 0.05 - 0.7 % of those adds, cause never
// established -- tools/stress_stats.py replays that path from a diagnostic build.)  Cost: the rows of the ~11 % of
// bins that finish in a lock-step are touched, 2 MB for 65 536 bins, resident in L2 / the Infinity Cache.
__device__ __forceinline__ void episode_acc_add(double *ep_acc, int e, double ret, double ratio, int len) {
    double *a = (double *)__builtin_assume_aligned(ep_acc + 4 * (size_t)e, 32);
    const double v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3];
    a[0] = v0 + ret;
    a[1] = v1 + ratio;
    a[2] = v2 + (double)len;
    a[3] = v3 + 1.0;
}
// Accumulators that every workgroup of a launch adds to: system scope, i.e. at the memory side.
__device__ __forceinline__ void stat_add_shared(double *p, double v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Per-bin record in LDS written by the bin's lane, read by the cell lanes.
struct __attribute__((aligned(16))) BinRec {
    uint32_t item;   // item shown in the next observation: x | y<<8 | z<<16
    uint32_t place;  // lx | ly<<8 | x<<16 | y<<24 of the box just placed
    uint32_t flags;  // bit0 placed, bit1 reset (zero the map), bits 8.. new top height
    uint32_t any;    // set to 1 by any feasible candidate
};

// Workgroups are dealt round-robin over the 8 XCDs, each with its own L2: block b runs on XCD (b + c) % 8, where the
// offset c is the same for all blocks of a launch but not always the same from launch to launch (tools/xcc_map.hip,
// profiles/archive/r03g_xcc_map.jsonl: 7 for a process's first launch, 6 afterwards).  Give every XCD one contiguous eighth of
// the bins so that cache lines shared by neighbouring waves (the small per-bin outputs, the byte heightmaps) are
// completed inside ONE L2 instead of being written back as partial lines from several.  Bijective for any grid size;
// affects speed only.
__device__ __forceinline__ int xcd_block_of(int b, int nb, int remap) {
    if (!remap || nb < 16) return b;
    const int xcd = b & 7, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}
__device__ __forceinline__ int xcd_block(int remap) { return xcd_block_of((int)blockIdx.x, (int)gridDim.x, remap); }

// Caller-supplied item sizes (mask-only entry points): each side is clamped into a byte so that it cannot
// spill into its neighbour's field; a side above 255 is wider than any supported bin either way.
__device__ __forceinline__ uint32_t pack_item(int x, int y, int z) {
    return (uint32_t)min(max(x, 0), 255) | ((uint32_t)min(max(y, 0), 255) << 8) | ((uint32_t)min(max(z, 0), 255) << 16);
}

__device__ __forceinline__ void wave_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving LDS
    // accesses across the point where other lanes' data is consumed.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (max, #cells == max, #corners == max, #corners == corner-max) of window [lx,lx+x) x [ly,ly+y).
// acktr/utils.py:14-16,23-26 and envs/bpp0/space.py:117-129.
struct Win {
    int mh, ma, c, sc;
};
__device__ __forceinline__ Win scan_window(const uint8_t *hm, int L, int lx, int ly, int x, int y) {
    const uint8_t *p = hm + lx * L + ly;
    int mh = 0, ma = 0;
    for (int a = 0; a < x; ++a) {
        const uint8_t *row = p + a * L;
        for (int b = 0; b < y; ++b) {
            int v = row[b];
            ma = v > mh ? 1 : ma + (v == mh);
            mh = v > mh ? v : mh;
        }
    }
    int r00 = p[0], r10 = p[(x - 1) * L], r01 = p[y - 1], r11 = p[(x - 1) * L + y - 1];
    int rm = max(max(r00, r10), max(r01, r11));
    Win w;
    w.mh = mh;
    w.ma = ma;
    w.c = (r00 == mh) + (r10 == mh) + (r01 == mh) + (r11 == mh);
    w.sc = (r00 == rm) + (r10 == rm) + (r01 == rm) + (r11 == rm);
    return w;
}

// Integer form of the float64 tests `max_area/area > 0.95 / 0.85 / 0.50` (exhaustively
// equal for 1 <= max_area <= area <= 1600; tests/test_host_logic.py::test_threshold_integer_rewrite_is_exact
// re-proves it for every window the kernels accept).
// Rule U: acktr/utils.py:20-33.  Rule S (envs/bpp0/space.py:122-142) == rule U && sc >= 3, because when
// the corner maximum rm equals max_h the two corner counts coincide and otherwise c == 0 in rule U.
__device__ __forceinline__ bool feasible(const Win &w, int area, int z, int H, int rule) {
    bool ok = (w.mh + z <= H) &&
              ((20 * w.ma > 19 * area) || (w.c == 3 && 20 * w.ma > 17 * area) || (w.c == 4 && 2 * w.ma > area));
    if (rule == BPP_RULE_SPACE) ok = ok && (w.sc >= 3);
    return ok;
}

// Counter-based RNG of the benchmark/test action sources (include/bpp_abi.h): 32-bit multiply-xorshift hash
// of (seed, global bin id, step).  The seed/step part is wave-uniform and lives on the scalar unit.
__device__ __forceinline__ uint32_t mix32_base(uint64_t seed, uint64_t step) {
    return ((uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B1u)) ^
           (((uint32_t)step + (uint32_t)(step >> 32) * 0xC2B2AE3Du) * 0x27D4EB2Fu);
}
__device__ __forceinline__ uint32_t mix32(uint32_t base, uint32_t gid) {
    uint32_t h = base ^ (gid * 0x85EBCA77u);
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

#include "bpp_rt_kernels.inl"
#include "bpp_tile_kernel.inl"
#include "bpp_stream_gen.inl"

#include "bpp_heads.inl"
#include "bpp_stats.inl"

// launch shape of the compaction: at most kGatherMaxGroups workgroups, chunks of whole rounds
static inline void gather_shape(int E, int &groups, int &chunk) {
    const int rounds = (E + kGatherRound - 1) / kGatherRound;
    const int want = rounds < kGatherMaxGroups ? rounds : kGatherMaxGroups;
    chunk = (rounds + want - 1) / want * kGatherRound;
    groups = (E + chunk - 1) / chunk;
}

thread_local char g_err[256];

int fail(int code, const char *msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

int check_geometry(int E, int W, int L, int H, int rotation, int rule) {
    if (E <= 0 || W <= 0 || L <= 0 || H <= 0) return fail(BPP_E_BADARG, "non-positive size");
    if (rotation != 0 && rotation != 1) return fail(BPP_E_BADARG, "rotation must be 0 or 1");
    if (rule != BPP_RULE_UTILS && rule != BPP_RULE_SPACE) return fail(BPP_E_BADARG, "unknown mask rule");
    if (W > kMaxDim || L > kMaxDim || H > kMaxDim || W * L > kMaxArea)
        return fail(BPP_E_TOOLARGE, "bin too large: need W,L,H <= 255 and W*L <= 1024");
    return 0;
}

// Geometry-dependent launch configuration.  EPW = bins per wave: as many as keep a 4-wave block's
// LDS under ~32 KiB (>= 5 blocks = 20 waves per CU), at most 16.
struct Launch {
    Params p;
    bool vec;
    int fast;  // kRuntimeGeo + K - 1: prefix-image kernel with runtime geometry, -1 = generic kernel
    int tile;  // index into kTileGeo (compile-time geometry, default launch shape), -1 = not the tile kernel
    int nit;   // tile kernel: groups of bins a wave walks through (1, 2 or 4)
    int wpb;   // waves per workgroup (waves are independent; this only sets the LDS/dispatch granule)
    int blocks;
    size_t lds;
};

// Geometries with a compiled tile kernel: (W, L, K, EPW) with K 64-bit histogram words, H + 2 <= 12 * K, EPW bins
// per wave.  Any other bin with W*L % 4 == 0 and H <= 22 -- and these, when the launch-shape knobs are set --
// runs bpp_fast_kernel with runtime geometry.
struct TileGeoEntry {
    int W, L, K, epw, nit, nit_big, nit_big_rot;   // groups per wave of the step kernel: default / when the outputs of one launch
};                                                 // exceed the 256 MiB Infinity Cache, without / with rotation (measured)
// Round 6 (profiles/r6e_sweep_bins_by_tile_groups_*.txt, r6k_*; four boxes): with ONE group per wave a launch of 262 144 / 1 048 576 10x10
// bins costs 14 - 17 % more per bin than a 65 536-bin launch (146 - 152 us instead of 4 x 31.8; eight and more rounds of workgroups:
// the later rounds' cold reads queue behind the earlier rounds' write streams), with two or four groups per wave it does not (128 -
// 133 / 124 - 136 us; 1 048 576 bins: 477 - 515 / 466 - 532 us = 2.0 - 2.25 G env steps/s) -- fewer, longer workgroups whose loads all
// go out before any of their stores.  Two and four groups are within +- 3 % of each other, box by box and run by run: two stay
// (with rotation two win clearly: 155 vs 162 us with four, 175 with one).
constexpr TileGeoEntry kTileGeo[] = {{10, 10, 1, 4, 1, 2, 2}, {20, 20, 1, 1, 1, 4, 4}, {20, 20, 2, 1, 1, 4, 4}, {10, 10, 2, 4, 1, 2, 2}};
constexpr size_t kOutputsPastL3 = 300u * 1000u * 1000u;   // obs + mask bytes per launch
constexpr int kNumTileGeo = sizeof(kTileGeo) / sizeof(kTileGeo[0]);
constexpr int kRuntimeGeo = 100;  // l.fast == kRuntimeGeo (K = 1) or kRuntimeGeo + 1 (K = 2)

// Tuning knobs (include/bpp_abi.h: bpp_knobs).  Initialised ONCE per process from the environment
// (BPP_EPW, BPP_WPB, BPP_XCD, BPP_FORCE_GENERIC, BPP_ABLATE), afterwards only bpp_set_knobs changes them:
// a launch never looks at the environment.
std::mutex g_knob_mutex;
bpp_knobs g_knobs;
bool g_knobs_init = false;

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

bpp_knobs current_knobs() {
    std::lock_guard<std::mutex> lock(g_knob_mutex);
    if (!g_knobs_init) {
        g_knobs.bins_per_wave = env_int("BPP_EPW", 0);
        g_knobs.waves_per_group = env_int("BPP_WPB", 0);
        g_knobs.xcd_remap = env_int("BPP_XCD", 1);
        g_knobs.force_generic = env_int("BPP_FORCE_GENERIC", 0);
        g_knobs.ablate = env_int("BPP_ABLATE", 0);
        g_knobs.legacy_fast = env_int("BPP_LEGACY_FAST", 0);
        g_knobs.tile_groups = env_int("BPP_TILE_GROUPS", 0);
        g_knobs.stream_legacy = env_int("BPP_STREAM_LEGACY", 0);
        g_knobs.stream_overlap = env_int("BPP_STREAM_OVERLAP", 1);
        g_knobs_init = true;
    }
    return g_knobs;
}

Launch configure(int E, int W, int L, int H, int rotation, int rule) {
    Launch l;
    Params &p = l.p;
    memset(&p, 0, sizeof p);
    p.E = E;
    p.W = W;
    p.L = L;
    p.H = H;
    p.A = W * L;
    p.rotation = rotation;
    p.M = p.A * (1 + rotation);
    p.rule = rule;
    l.vec = (p.A % 4) == 0;
    const bpp_knobs kn = current_knobs();
    int epw = 16;
    if (kn.bins_per_wave > 0) epw = kn.bins_per_wave > 64 ? 64 : kn.bins_per_wave;
    else
        while (epw > 1 && (size_t)kWavesPerBlock * (epw * (p.A + p.M + 16)) > 32 * 1024) epw >>= 1;
    l.fast = -1;
    l.tile = -1;
    l.nit = 1;
    const bool gen = kn.force_generic != 0;
    if (!gen && !kn.legacy_fast && kn.bins_per_wave <= 0 && kn.waves_per_group <= 0)
        for (int g = 0; g < kNumTileGeo; ++g)
            if (kTileGeo[g].W == W && kTileGeo[g].L == L && H + 2 <= kLevelsPerWord * kTileGeo[g].K) {
                l.tile = g;
                break;
            }
    // any other bin whose area is a multiple of 4 and whose heights fit two histogram words runs the same
    // algorithm with runtime geometry (kRuntimeGeo + K - 1)
    int rt_k = 0;
    if (!gen && l.fast < 0 && l.vec && H + 2 <= kLevelsPerWord * 2) {
        rt_k = H + 2 <= kLevelsPerWord ? 1 : 2;
        l.fast = kRuntimeGeo + rt_k - 1;
    }
    const int pn_bytes = l.fast >= 0 ? (W + 1) * (L + 1) * 8 * rt_k : 0;
    if (l.fast >= 0 && kn.bins_per_wave <= 0) {
        // prefix image dominates LDS: keep a 4-wave block under 24 KiB (>= 6 blocks = 24 waves per CU).
        // Measured on MI355X: 10x10: EPW=4 39 us vs 43 us at EPW=8 and 50 us at EPW=2; 10x10 + rotation
        // (21 KiB at EPW=4): 51 us vs 59 us at EPW=2; 20x20: EPW=1 85 us vs 116 us at EPW=2.
        epw = 16;
        while (epw > 1 && (size_t)kWavesPerBlock * (epw * (p.A + p.M + 48 + pn_bytes)) > 24 * 1024) epw >>= 1;
    }
    if (l.fast >= 0) {  // sub-groups of 64/epw lanes per bin: epw must be a power of two <= 64
        int sh = 0;
        while ((2 << sh) <= epw && sh < 6) ++sh;
        epw = 1 << sh;
        p.epw_shift = sh;
    }
    p.xcd_remap = kn.xcd_remap;
    p.ablate = kn.ablate;
    p.epw = epw;
    p.off_mk = (epw * p.A + 15) & ~15;
    p.off_rec = (p.off_mk + epw * p.M + 15) & ~15;
    p.off_ori = p.off_rec + epw * (int)sizeof(BinRec);
    p.off_P = p.off_ori + (l.fast >= 0 ? epw * 2 * (int)sizeof(OriRec) : 0);
    p.lds_per_wave = p.off_P + epw * pn_bytes;
    p.divL = make_fastdiv(L);
    p.divA = make_fastdiv(p.A);
    p.divM = make_fastdiv(p.M);
    p.divA4 = make_fastdiv(l.vec ? p.A / 4 : p.A);
    p.divW = make_fastdiv(W);
    p.divM4 = make_fastdiv(p.M / 4 > 0 ? p.M / 4 : 1);
    p.divPWW = make_fastdiv(L + 1 + W);
    p.binvol = (double)W * (double)L * (double)H;
    const int waves = (E + epw - 1) / epw;
    l.wpb = kWavesPerBlock;
    if (kn.waves_per_group >= 1 && kn.waves_per_group <= (l.fast >= 0 ? kMaxFastWavesPerBlock : kWavesPerBlock))
        l.wpb = kn.waves_per_group;
    if (l.fast >= 0 && l.wpb * epw > kWave) l.wpb = kWave / epw;  // wave 0 carries one bin per lane
    l.blocks = (waves + l.wpb - 1) / l.wpb;
    l.lds = (size_t)l.wpb * p.lds_per_wave;
    if (l.tile >= 0) {   // the tile kernel's launch shape is part of its type; only the grid depends on E
        const bool past_l3 = (size_t)E * (size_t)(16 * p.A + 4 * p.M) > kOutputsPastL3;
        l.nit = (kn.tile_groups == 1 || kn.tile_groups == 2 || kn.tile_groups == 4)
                    ? kn.tile_groups : (past_l3 ? (rotation ? kTileGeo[l.tile].nit_big_rot : kTileGeo[l.tile].nit_big) : kTileGeo[l.tile].nit);
        const int nb = kTileWaves * kTileGeo[l.tile].epw * l.nit;   // step kernel; reset / mask kernels: one group
        p.epw = kTileGeo[l.tile].epw;
        l.wpb = kTileWaves;
        l.blocks = (E + nb - 1) / nb;
        l.lds = 0;       // taken from TileGeo<...>::LDS_BLOCK at launch
    }
    return l;
}

template <int W, int L, int K, bool ROT, int MODE>
void launch_fast_rot(const Launch &l, hipStream_t s) {
    auto kern = bpp_fast_kernel<W, L, K, ROT, MODE>;
    if (l.lds > 64 * 1024) {  // large workgroups: opt in to more than 64 KiB of dynamic LDS, once per kernel AND device
        static std::atomic<uint64_t> raised{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t bit = 1ull << (dev & 63);
        if (!(raised.load(std::memory_order_acquire) & bit)) {
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(kern, dim3(l.blocks), dim3(kWave * l.wpb), l.lds, s, l.p);
}

template <int W, int L, int K, int MODE>
void launch_fast(const Launch &l, hipStream_t s) {
    if (l.p.rotation)
        launch_fast_rot<W, L, K, true, MODE>(l, s);
    else
        launch_fast_rot<W, L, K, false, MODE>(l, s);
}

template <int W, int L, int K, int MODE, int EPW, int NIT>
void launch_tile_nit(const Launch &l, hipStream_t s) {
    const int nb = kTileWaves * EPW * NIT;
    const int blocks = (l.p.E + nb - 1) / nb;
    if (MODE == kStep && l.p.cache != nullptr) {   // row cache: copier workgroups (kCopierBins bins each) in front of the grid
        Params q = l.p;
        q.ncopy = (l.p.E + kCopierBins - 1) / kCopierBins;
        if (l.p.rotation)
            hipLaunchKernelGGL((bpp_tile_kernel_q<W, L, K, true, kStep, EPW, NIT>), dim3(blocks + q.ncopy), dim3(kWave * kTileWaves),
                               (TileGeo<W, L, K, true, EPW, NIT>::LDS_BLOCK), s, q);
        else
            hipLaunchKernelGGL((bpp_tile_kernel_q<W, L, K, false, kStep, EPW, NIT>), dim3(blocks + q.ncopy), dim3(kWave * kTileWaves),
                               (TileGeo<W, L, K, false, EPW, NIT>::LDS_BLOCK), s, q);
    } else if (MODE == kMaskObs || MODE == kMaskHmap) {
        // The mask-only entry points have no deciding wave and no workgroup barrier: their waves are launched as workgroups of ONE
        // (round 6: finer-grained dispatch, waves retire and start one by one instead of in fours: 21.8 / 20.7 -> 21.1 / 20.2 us, with
        // rotation 29.7 / 28.5 -> 28.7 / 27.7, 20x20x20 43.6 / 42.1 -> 41.5 / 39.4; two waves per workgroup: no gain)
        const int mblocks = (l.p.E + EPW - 1) / EPW;
        if (l.p.rotation)
            hipLaunchKernelGGL((bpp_tile_kernel<W, L, K, true, MODE, EPW, NIT>), dim3(mblocks), dim3(kWave),
                               (size_t)(TileGeo<W, L, K, true, EPW, NIT>::LDS_WAVE), s, l.p);
        else
            hipLaunchKernelGGL((bpp_tile_kernel<W, L, K, false, MODE, EPW, NIT>), dim3(mblocks), dim3(kWave),
                               (size_t)(TileGeo<W, L, K, false, EPW, NIT>::LDS_WAVE), s, l.p);
    } else if (l.p.rotation)
        hipLaunchKernelGGL((bpp_tile_kernel<W, L, K, true, MODE, EPW, NIT>), dim3(blocks), dim3(kWave * kTileWaves),
                           (TileGeo<W, L, K, true, EPW, NIT>::LDS_BLOCK), s, l.p);
    else
        hipLaunchKernelGGL((bpp_tile_kernel<W, L, K, false, MODE, EPW, NIT>), dim3(blocks), dim3(kWave * kTileWaves),
                           (TileGeo<W, L, K, false, EPW, NIT>::LDS_BLOCK), s, l.p);
}

// The step kernel is compiled for 1, 2 and 4 groups per wave (l.nit); reset and the mask-only entry points have no
// per-bin chain worth amortising and always run one group per wave.
template <int W, int L, int K, int MODE, int EPW>
void launch_tile(const Launch &l, hipStream_t s) {
    if (MODE == kStep && l.nit == 4)
        launch_tile_nit<W, L, K, MODE, EPW, MODE == kStep ? 4 : 1>(l, s);
    else if (MODE == kStep && l.nit == 2)
        launch_tile_nit<W, L, K, MODE, EPW, MODE == kStep ? 2 : 1>(l, s);
    else
        launch_tile_nit<W, L, K, MODE, EPW, 1>(l, s);
}

template <int MODE>
int launch(const Launch &l, hipStream_t s) {
    if (l.lds > (l.fast >= 0 ? 160 : 64) * 1024) return fail(BPP_E_TOOLARGE, "LDS request per workgroup too large");
    if (l.tile == 0)
        launch_tile<10, 10, 1, MODE, 4>(l, s);
    else if (l.tile == 1)
        launch_tile<20, 20, 1, MODE, 1>(l, s);
    else if (l.tile == 2)
        launch_tile<20, 20, 2, MODE, 1>(l, s);
    else if (l.tile == 3)
        launch_tile<10, 10, 2, MODE, 4>(l, s);
    else if (l.fast == kRuntimeGeo)
        launch_fast<0, 0, 1, MODE>(l, s);
    else if (l.fast == kRuntimeGeo + 1)
        launch_fast<0, 0, 2, MODE>(l, s);
    else if (l.vec)
        hipLaunchKernelGGL((bpp_kernel<true, MODE>), dim3(l.blocks), dim3(kWave * l.wpb), l.lds, s, l.p);
    else
        hipLaunchKernelGGL((bpp_kernel<false, MODE>), dim3(l.blocks), dim3(kWave * l.wpb), l.lds, s, l.p);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

int fill_batch(Launch &l, const bpp_batch *b, const bpp_step_out *out, bool need_all) {
    if (!b->seq_pool || !b->hmap || !b->state) return fail(BPP_E_BADARG, "bpp_batch: NULL pointer");
    if (b->pool_size <= 0 || b->pool_len <= 0) return fail(BPP_E_BADARG, "bpp_batch: empty pool");
    if (b->env_id_base < 0 || b->env_id_total < b->env_id_base + b->num_envs)
        return fail(BPP_E_BADARG, "bpp_batch: env_id_total < env_id_base + num_envs");
    if (!out || !out->obs) return fail(BPP_E_BADARG, "bpp_step_out: NULL obs");
    if (need_all && (!out->reward || !out->done || !out->counter || !out->ratio || !out->ep_ret || !out->ep_len))
        return fail(BPP_E_BADARG, "bpp_step_out: NULL pointer");
    if ((out->host_reward == nullptr) != (out->host_done == nullptr))
        return fail(BPP_E_BADARG, "bpp_step_out: host_reward and host_done go together");
    if (((uintptr_t)b->hmap & 3u) || !aligned16(b->state) || !aligned16(out->obs) || (out->mask && !aligned16(out->mask)) ||
        ((uintptr_t)b->seq_pool & 3u))
        return fail(BPP_E_BADARG, "buffers must be 16-byte aligned");
    if ((uintptr_t)b->ep_acc & 31u) return fail(BPP_E_BADARG, "bpp_batch: ep_acc must be 32-byte aligned");
    Params &p = l.p;
    p.P = b->pool_size;
    p.T = b->pool_len;
    if (b->pool_mode == BPP_POOL_RING) {   // ring of a bpp_stream: row = (episode mod depth) * num_envs + local bin
        if (b->pool_size % b->num_envs != 0 || b->pool_size / b->num_envs < 4)
            return fail(BPP_E_BADARG, "bpp_batch: a ring pool holds depth * num_envs rows, depth >= 4");
        if (b->pool_len < 4) return fail(BPP_E_BADARG, "bpp_batch: ring rows hold two look-ahead entries, at least one item and the terminator");
        p.seq_stride = b->num_envs % b->pool_size;
        p.ring2 = 2;
        p.base_mod = 0;
        if (b->seq_cache != nullptr) {
            if ((uintptr_t)b->seq_cache & 127u) return fail(BPP_E_BADARG, "bpp_batch: seq_cache must be 128-byte aligned");
            if (b->pool_size / b->num_envs < 5) return fail(BPP_E_BADARG, "bpp_batch: seq_cache needs a ring of depth >= 5");
            if (b->pool_len > 0x1fff) return fail(BPP_E_BADARG, "bpp_batch: seq_cache needs pool_len < 8192");
            p.cache = (unsigned char *)b->seq_cache;
        }
    } else if (b->pool_mode == BPP_POOL_STATIC) {
        p.seq_stride = (int32_t)(b->env_id_total % b->pool_size);
        p.ring2 = 0;
        p.base_mod = (int32_t)(b->env_id_base % b->pool_size);
        if (b->seq_cache != nullptr) return fail(BPP_E_BADARG, "bpp_batch: seq_cache goes with BPP_POOL_RING");
    } else {
        return fail(BPP_E_BADARG, "bpp_batch: unknown pool_mode");
    }
    p.pool = (const uint32_t *)b->seq_pool;
    p.hmap = b->hmap;
    p.state = b->state;
    p.ep_acc = b->ep_acc;
    p.obs = out->obs;
    p.mask = out->mask;
    p.reward = out->reward;
    p.done = out->done;
    p.host_reward = out->host_reward;
    p.host_done = out->host_done;
    p.counter = out->counter;
    p.ratio = out->ratio;
    p.ep_ret = out->ep_ret;
    p.ep_len = out->ep_len;
    p.next_action = out->next_action;
    p.sample_seed = out->sample_seed;
    p.sample_step = out->sample_step;
    p.env_id_base = b->env_id_base;
    return 0;
}

// bpp_epsilon_override (include/bpp_abi.h): with probability eps_q24 / 2^24 a bin's action becomes a uniform draw over ALL M
// entries -- SURVEY 8d's failure-path variant of the benchmark policy.  One thread per bin.
__global__ void eps_override_kernel(int64_t *actions, int E, int M, int64_t env_id_base, uint64_t seed, uint64_t step, uint32_t eps_q24) {
    const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= E) return;
    const uint32_t gid = (uint32_t)(env_id_base + e);
    const uint32_t h = mix32(mix32_base(seed ^ BPP_EPS_KEY_COIN, step), gid);
    if ((h >> 8) < eps_q24) actions[e] = (int64_t)__umulhi(mix32(mix32_base(seed ^ BPP_EPS_KEY_PICK, step), gid), (uint32_t)M);
}

}  // namespace

extern "C" {

int bpp_abi_version(void) { return BPP_ABI_VERSION; }

const char *bpp_last_error(void) { return g_err; }

int bpp_get_knobs(bpp_knobs *out) {
    if (!out) return fail(BPP_E_BADARG, "bpp_get_knobs: NULL");
    *out = current_knobs();
    return 0;
}

int bpp_set_knobs(const bpp_knobs *k) {
    if (!k) return fail(BPP_E_BADARG, "bpp_set_knobs: NULL");
    if (k->bins_per_wave < 0 || k->bins_per_wave > 64 || k->waves_per_group < 0 || k->waves_per_group > kMaxFastWavesPerBlock)
        return fail(BPP_E_BADARG, "bpp_set_knobs: bins_per_wave must be 0..64, waves_per_group 0..16");
    (void)current_knobs();
    std::lock_guard<std::mutex> lock(g_knob_mutex);
    g_knobs = *k;
    return 0;
}

#ifdef BPP_ENABLE_ABLATION
// profiling builds only: copy the phase timestamps to the host (n = number of uint64 values, <= 4096 * 16)
int bpp_debug_stamps(unsigned long long *host_out, int n, int clear) {
    if (!host_out || n <= 0 || n > kStampWaves * kStampSlots) return fail(BPP_E_BADARG, "bpp_debug_stamps: bad argument");
    hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyFromSymbol");
    if (clear) {
        std::vector<unsigned long long> z((size_t)kStampWaves * kStampSlots, 0ull);
        e = hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z.data(), z.size() * 8, 0, hipMemcpyHostToDevice);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyToSymbol");
    }
    return 0;
}
#endif

int bpp_launch_info(int32_t E, int32_t W, int32_t L, int32_t H, int32_t rotation, int32_t out[6]) {
    if (!out) return fail(BPP_E_BADARG, "bpp_launch_info: NULL");
    int rc = check_geometry(E, W, L, H, rotation, BPP_RULE_UTILS);
    if (rc) return rc;
    const Launch l = configure(E, W, L, H, rotation, BPP_RULE_UTILS);
    out[0] = l.tile >= 0 ? BPP_KERNEL_TILE : (l.fast >= 0 ? BPP_KERNEL_PREFIX_RT : BPP_KERNEL_CELLSCAN);
    out[1] = l.tile >= 0 ? kTileGeo[l.tile].K : (l.fast >= 0 ? l.fast - kRuntimeGeo + 1 : 0);
    out[2] = l.p.epw;
    out[3] = l.wpb;
    out[4] = l.blocks;
    size_t lds = l.lds;
    if (l.tile >= 0) {   // step kernel shape (TileGeo<...>::LDS_BLOCK restated for runtime arguments)
        const TileGeoEntry &g = kTileGeo[l.tile];
        const int A = W * L, M = A * (1 + rotation), npass = (A + kWave - 1) / kWave, nbw = g.epw * l.nit;
        const int off_mk = round16(nbw * A), off_rec = round16(off_mk + g.epw * M);
        const int off_bal = (off_rec + nbw * (int)sizeof(TileRec) + 7) & ~7;
        const int off_p = round16(off_bal + (npass > 2 ? g.epw * 2 * npass * 8 : 0));
        lds = (size_t)kTileWaves * (off_p + g.epw * (W + 1) * (L + 1) * 8 * (g.epw == 1 ? 1 : g.K));   // TileGeo::KP
        out[2] = nbw;
    }
    out[5] = (int32_t)lds;
    return 0;
}

int bpp_limits(int32_t out[2]) {
    if (!out) return fail(BPP_E_BADARG, "bpp_limits: NULL");
    out[0] = kMaxArea;
    out[1] = kMaxDim;
    return 0;
}

int bpp_reset(const bpp_batch *b, int32_t mode, const bpp_step_out *out, void *stream) {
    if (!b) return fail(BPP_E_BADARG, "bpp_reset: NULL batch");
    if (mode != BPP_RESET_INIT && mode != BPP_RESET_ADVANCE) return fail(BPP_E_BADARG, "bpp_reset: bad mode");
    int rc = check_geometry(b->num_envs, b->W, b->L, b->H, b->rotation, b->mask_rule);
    if (rc) return rc;
    Launch l = configure(b->num_envs, b->W, b->L, b->H, b->rotation, b->mask_rule);
    rc = fill_batch(l, b, out, false);
    if (rc) return rc;
    return mode == BPP_RESET_INIT ? launch<kResetInit>(l, (hipStream_t)stream)
                                  : launch<kResetAdvance>(l, (hipStream_t)stream);
}

int bpp_step(const bpp_batch *b, const int64_t *actions, const bpp_step_out *out, void *stream) {
    if (!b || !actions) return fail(BPP_E_BADARG, "bpp_step: NULL pointer");
    int rc = check_geometry(b->num_envs, b->W, b->L, b->H, b->rotation, b->mask_rule);
    if (rc) return rc;
    Launch l = configure(b->num_envs, b->W, b->L, b->H, b->rotation, b->mask_rule);
    rc = fill_batch(l, b, out, true);
    if (rc) return rc;
    if (out->next_action && !out->mask) return fail(BPP_E_BADARG, "bpp_step: next_action needs mask");
    l.p.actions = actions;
    if (l.fast < 0) l.p.next_action = nullptr;  // the generic kernel has no fused sampler ...
    rc = launch<kStep>(l, (hipStream_t)stream);
    if (rc == 0 && l.fast < 0 && out->next_action)  // ... a separate launch draws from the mask it wrote
        rc = bpp_sample_feasible(out->mask, out->next_action, b->num_envs, b->W * b->L * (1 + b->rotation),
                                 b->env_id_base, out->sample_seed, out->sample_step, stream);
    return rc;
}

int bpp_mask_from_obs(const float *obs, float *mask, int32_t E, int32_t W, int32_t L, int32_t H, int32_t rotation,
                      int32_t rule, void *stream) {
    if (!obs || !mask) return fail(BPP_E_BADARG, "bpp_mask_from_obs: NULL pointer");
    int rc = check_geometry(E, W, L, H, rotation, rule);
    if (rc) return rc;
    if (!aligned16(obs) || !aligned16(mask)) return fail(BPP_E_BADARG, "buffers must be 16-byte aligned");
    Launch l = configure(E, W, L, H, rotation, rule);
    l.p.obs_in = obs;
    l.p.mask = mask;
    return launch<kMaskObs>(l, (hipStream_t)stream);
}

int bpp_mask_from_hmap(const int32_t *hmap, const int32_t *items, float *mask, int32_t E, int32_t W, int32_t L,
                       int32_t H, int32_t rotation, int32_t rule, void *stream) {
    if (!hmap || !items || !mask) return fail(BPP_E_BADARG, "bpp_mask_from_hmap: NULL pointer");
    int rc = check_geometry(E, W, L, H, rotation, rule);
    if (rc) return rc;
    if (!aligned16(hmap) || !aligned16(mask)) return fail(BPP_E_BADARG, "buffers must be 16-byte aligned");
    Launch l = configure(E, W, L, H, rotation, rule);
    l.p.hmap_in = hmap;
    l.p.items_in = items;
    l.p.mask = mask;
    return launch<kMaskHmap>(l, (hipStream_t)stream);
}

int bpp_sample_feasible(const float *mask, int64_t *actions, int32_t E, int32_t M, int64_t env_id_base, uint64_t seed,
                        uint64_t step, void *stream) {
    if (!mask || !actions) return fail(BPP_E_BADARG, "bpp_sample_feasible: NULL pointer");
    if (E <= 0 || M <= 0) return fail(BPP_E_BADARG, "bpp_sample_feasible: non-positive size");
    hipStream_t st = (hipStream_t)stream;
    const int nq = M / 4;
    const int per = (nq + 15) / 16;
    if (M % 4 == 0 && per <= 8 && (((uintptr_t)mask) & 15u) == 0) {
        const int blocks = (E + 15) / 16;  // 16 bins per 256-thread block
#define BPP_SAMPLE(P) hipLaunchKernelGGL(sample_kernel<P>, dim3(blocks), dim3(256), 0, st, mask, actions, E, M, env_id_base, seed, step)
        switch (per) {
            case 1: BPP_SAMPLE(1); break;
            case 2: BPP_SAMPLE(2); break;
            case 3: BPP_SAMPLE(3); break;
            case 4: BPP_SAMPLE(4); break;
            case 5: case 6: BPP_SAMPLE(6); break;
            default: BPP_SAMPLE(8); break;
        }
#undef BPP_SAMPLE
    } else {
        hipLaunchKernelGGL(sample_kernel_generic, dim3((E + 3) / 4), dim3(256), 0, st, mask, actions, E, M, env_id_base,
                           seed, step);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

int bpp_epsilon_override(int64_t *actions, int32_t E, int32_t M, int64_t env_id_base, uint64_t seed, uint64_t step, uint32_t eps_q24,
                         void *stream) {
    if (!actions) return fail(BPP_E_BADARG, "bpp_epsilon_override: NULL pointer");
    if (E <= 0 || M <= 0 || eps_q24 > (1u << 24)) return fail(BPP_E_BADARG, "bpp_epsilon_override: bad size / eps_q24 > 2^24");
    if (eps_q24 == 0) return 0;
    hipLaunchKernelGGL(eps_override_kernel, dim3((E + 255) / 256), dim3(256), 0, (hipStream_t)stream, actions, E, M, env_id_base, seed,
                       step, eps_q24);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

}  // extern "C" (reopened below: a template cannot have C linkage)

namespace {

// Rows [0, n) split over host threads; fn(k0, k1) returns a status, the maximum is returned.
template <typename F>
int run_rows_threaded(int n, int threads, F fn) {
    int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 64 ? 64 : nt);
    if (nt > n) nt = n;
    std::vector<int> status((size_t)nt, 0);
    std::vector<std::thread> workers;
    for (int t = 0; t < nt; ++t) {
        const int k0 = (int)((int64_t)n * t / nt), k1 = (int)((int64_t)n * (t + 1) / nt);
        workers.emplace_back([=, &status] { status[(size_t)t] = fn(k0, k1); });
    }
    for (auto &th : workers) th.join();
    int worst = 0;
    for (int v : status) worst = v > worst ? v : worst;
    return worst;
}

}  // namespace

extern "C" {

int bpp_gen_cut2(uint8_t *pool, int32_t *lengths, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H, int32_t bound_lo,
                 int32_t bound_hi, uint64_t seed0, int32_t threads) {
    if (!pool || !bpp_gen_cut2_args_ok(n, T, W, L, H, bound_lo, bound_hi))
        return fail(BPP_E_BADARG, "bpp_gen_cut2: bad argument (bin must exceed bound_hi on some side and bound_lo on none)");
    // same generator as the device stream (bpp_stream_gen.inl), one private random.Random(seed0 + k) per row
    const int st = run_rows_threaded(n, threads, [=](int k0, int k1) {
        std::vector<uint32_t> mt(624);
        const int maxn = stream_work_entries(W, L, H, bound_lo);
        std::vector<CutBox> boxes((size_t)maxn);
        std::vector<uint32_t> cut((size_t)maxn);
        const uint32_t term = (uint32_t)W | ((uint32_t)L << 8) | ((uint32_t)H << 16);
        int over = 0;
        for (int k = k0; k < k1; ++k) {
            StridedMT rng{mt.data(), 1, 624};
            rng.seed(seed0 + (uint64_t)k);
            ArrayWork work{boxes.data()};
            ArrayVals vals{cut.data()};
            uint32_t *row = (uint32_t *)pool + (size_t)k * T;
            const int cnt = cut2_generate(rng, work, vals, W, L, H, bound_lo, bound_hi);
            const int nw = cnt < T - 1 ? cnt : T - 1;
            for (int t = 0; t < nw; ++t) row[t] = cut[(size_t)t] & 0x00ffffffu;
            for (int t = nw; t < T; ++t) row[t] = term;
            if (lengths) lengths[k] = cnt;
            over |= cnt > T - 1;
        }
        return over;
    });
    return st ? fail(BPP_E_TOOLARGE, "bpp_gen_cut2: a sequence does not fit in T-1 entries") : 0;
}

int bpp_gen_cut1(uint8_t *pool, int32_t *lengths, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H,
                 const int32_t box_range[6], int32_t rotation, uint64_t seed0, int32_t threads) {
    if (!pool || !bpp_gen_cut1_args_ok(n, T, W, L, H, box_range) || seed0 + (uint64_t)n > (1ull << 32))
        return fail(BPP_E_BADARG, "bpp_gen_cut1: bad argument (need low >= 1, high >= 2*low - 1, bin >= low, seeds < 2^32)");
    int32_t rg[6];
    memcpy(rg, box_range, sizeof rg);
    const int st = run_rows_threaded(n, threads, [=](int k0, int k1) {
        return bpp_gen_cut1_range(pool, lengths, k0, k1, T, W, L, H, rg, rotation != 0, seed0);
    });
    if (st == 2) return fail(BPP_E_BADARG, "bpp_gen_cut1: a piece fell below the lower bound (the reference asserts here, cutCreator.py:74)");
    return st ? fail(BPP_E_TOOLARGE, "bpp_gen_cut1: a sequence does not fit in T-1 entries") : 0;
}

int bpp_gen_rs(uint8_t *pool, int32_t n, int32_t T, int32_t W, int32_t L, int32_t H, const int32_t *box_set, int32_t n_box,
               uint64_t seed0, int32_t threads) {
    if (!pool || !bpp_gen_rs_args_ok(n, T, W, L, H, box_set, n_box) || seed0 + (uint64_t)n > (1ull << 32))
        return fail(BPP_E_BADARG, "bpp_gen_rs: bad argument (item sides 1..255, seeds < 2^32)");
    run_rows_threaded(n, threads, [=](int k0, int k1) {
        bpp_gen_rs_range(pool, k0, k1, T, W, L, H, box_set, n_box, seed0);
        return 0;
    });
    return 0;
}

}  // extern "C"

namespace {
int masked_act_launch(const float *logits, const float *mask, int64_t *action, float *log_prob, int32_t E, int32_t M, int64_t env_id_base,
                      uint64_t seed, uint64_t step, const uint64_t *seed_step, int32_t deterministic, void *stream, const char *who) {
    if (!logits || !mask || !action) return fail(BPP_E_BADARG, who);
    if (E <= 0 || M <= 0) return fail(BPP_E_BADARG, who);
    hipStream_t st = (hipStream_t)stream;
    if (M % 4 != 0 || M > 16 * 8 * 4 || !aligned16(logits) || !aligned16(mask)) {  // wave-per-bin kernel: any M, any alignment
        hipLaunchKernelGGL(masked_act_kernel_generic, dim3((E + 3) / 4), dim3(256), 0, st, logits, mask, action, log_prob, E, M,
                           env_id_base, seed, step, deterministic, seed_step);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
    }
    const int per = (M / 4 + 15) / 16;
    const int blocks = (E + 15) / 16;
#define BPP_ACT(P)                                                                                                                     \
    do {                                                                                                                               \
        if (deterministic)                                                                                                             \
            hipLaunchKernelGGL((masked_act_kernel<P, true>), dim3(blocks), dim3(256), 0, st, logits, mask, action, log_prob, E, M, env_id_base, seed, step, seed_step); \
        else                                                                                                                           \
            hipLaunchKernelGGL((masked_act_kernel<P, false>), dim3(blocks), dim3(256), 0, st, logits, mask, action, log_prob, E, M, env_id_base, seed, step, seed_step); \
    } while (0)
    switch (per) {
        case 1: BPP_ACT(1); break;
        case 2: BPP_ACT(2); break;
        case 3: BPP_ACT(3); break;
        case 4: BPP_ACT(4); break;
        case 5: case 6: BPP_ACT(6); break;
        default: BPP_ACT(8); break;
    }
#undef BPP_ACT
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}
}  // namespace

extern "C" {

int bpp_masked_act(const float *logits, const float *mask, int64_t *action, float *log_prob, int32_t E, int32_t M,
                   int64_t env_id_base, uint64_t seed, uint64_t step, int32_t deterministic, void *stream) {
    return masked_act_launch(logits, mask, action, log_prob, E, M, env_id_base, seed, step, nullptr, deterministic, stream,
                             "bpp_masked_act: NULL pointer / non-positive size");
}

int bpp_masked_act_counter(const float *logits, const float *mask, int64_t *action, float *log_prob, int32_t E, int32_t M,
                           int64_t env_id_base, const uint64_t *seed_step, int32_t deterministic, void *stream) {
    if (!seed_step) return fail(BPP_E_BADARG, "bpp_masked_act_counter: NULL seed_step");
    return masked_act_launch(logits, mask, action, log_prob, E, M, env_id_base, 0, 0, seed_step, deterministic, stream,
                             "bpp_masked_act_counter: NULL pointer / non-positive size");
}

int bpp_masked_evaluate(const float *logits, const float *mask, const int64_t *action, float *log_prob, float *entropy,
                        float *bad_prob, int32_t E, int32_t M, void *stream) {
    if (!logits || !mask || !action || !log_prob || !entropy || !bad_prob) return fail(BPP_E_BADARG, "bpp_masked_evaluate: NULL pointer");
    if (E <= 0 || M <= 0) return fail(BPP_E_BADARG, "bpp_masked_evaluate: non-positive size");
    hipLaunchKernelGGL(masked_eval_fwd_kernel, dim3((E + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, mask, action, log_prob,
                       entropy, bad_prob, E, M);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

int bpp_masked_evaluate_backward(const float *logits, const float *mask, const int64_t *action, const float *g_log_prob,
                                 const float *g_entropy, const float *g_bad_prob, float *grad_logits, int32_t E, int32_t M,
                                 void *stream) {
    if (!logits || !mask || !action || !g_log_prob || !g_entropy || !g_bad_prob || !grad_logits)
        return fail(BPP_E_BADARG, "bpp_masked_evaluate_backward: NULL pointer");
    if (E <= 0 || M <= 0) return fail(BPP_E_BADARG, "bpp_masked_evaluate_backward: non-positive size");
    hipLaunchKernelGGL(masked_eval_bwd_kernel, dim3((E + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, mask, action,
                       g_log_prob, g_entropy, g_bad_prob, grad_logits, E, M);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

}  // extern "C"

namespace {
// bpp_rollout_uniform, and a chunk of bpp_rollout_uniform_stream: draw_first = the first action comes from the mask left by the
// previous reset / step (a standalone bpp_sample_feasible launch that reads the whole mask); every step then draws the next
// one itself (bpp_step_out.next_action, in place), the last one only with draw_last (then `actions` holds the draw for step
// step0 + nsteps and the next chunk needs no launch of its own for it).
int rollout_uniform_steps(const bpp_batch *b, const bpp_step_out *out, int64_t *actions, uint64_t seed, uint64_t step0, int32_t nsteps,
                          bool draw_first, bool draw_last, void *stream) {
    const int M = b->W * b->L * (1 + b->rotation);
    int rc = draw_first ? bpp_sample_feasible(out->mask, actions, b->num_envs, M, b->env_id_base, seed, step0, stream) : 0;
    for (int t = 0; rc == 0 && t < nsteps; ++t) {
        bpp_step_out o = *out;
        o.next_action = (t + 1 < nsteps || draw_last) ? actions : nullptr;
        o.sample_seed = seed;
        o.sample_step = step0 + (uint64_t)t + 1;
        rc = bpp_step(b, actions, &o, stream);
    }
    return rc;
}
}  // namespace

extern "C" {

int bpp_rollout_uniform(const bpp_batch *b, const bpp_step_out *out, int64_t *actions, uint64_t seed, uint64_t step0,
                        int32_t nsteps, void *stream) {
    if (!b || !out || !out->mask || !actions) return fail(BPP_E_BADARG, "bpp_rollout_uniform: NULL pointer");
    if (nsteps < 0) return fail(BPP_E_BADARG, "bpp_rollout_uniform: negative nsteps");
    if (nsteps == 0) return 0;
    return rollout_uniform_steps(b, out, actions, seed, step0, nsteps, true, false, stream);
}

int bpp_fetch_to_host(const void *device_src, void *host_dst, int64_t nbytes, void *stream) {
    if (!device_src || !host_dst || nbytes <= 0) return fail(BPP_E_BADARG, "bpp_fetch_to_host: NULL pointer / non-positive size");
    hipError_t e = hipMemcpyAsync(host_dst, device_src, (size_t)nbytes, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync");
    e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "hipStreamSynchronize");
}

int bpp_gather_finished(const uint8_t *done, const double *ep_ret, const double *ratio, const int32_t *ep_len,
                        const int32_t *counter, int32_t E, void *dev, void *host, int32_t n, void *stream) {
    if (!done || !ep_ret || !ratio || !ep_len || !counter || !host) return fail(BPP_E_BADARG, "bpp_gather_finished: NULL pointer");
    if (E <= 0 || n < BPP_GATHER_ENQUEUE_ONLY || n > E) return fail(BPP_E_BADARG, "bpp_gather_finished: bad size");
    if (n == BPP_GATHER_ENQUEUE_ONLY) {
        // the eager form: the compaction is only ENQUEUED behind the step (arrays laid out for E entries, straight into mapped host
        // memory); the caller's one synchronisation of the step covers it and the header then tells how many entries there are
        if (dev) return fail(BPP_E_BADARG, "bpp_gather_finished: BPP_GATHER_ENQUEUE_ONLY writes into mapped host memory (dev must be NULL)");
        if ((uintptr_t)host & 7u) return fail(BPP_E_BADARG, "bpp_gather_finished: buffers must be 8-byte aligned");
        int groups0, chunk0;
        gather_shape(E, groups0, chunk0);
        hipLaunchKernelGGL(compact_finished_kernel, dim3(groups0), dim3(kGatherThreads), 0, (hipStream_t)stream, done, ep_ret, ratio, ep_len, counter, E,
                           (unsigned char *)host, E, chunk0);
        hipError_t e0 = hipGetLastError();
        return e0 == hipSuccess ? 0 : hip_fail(e0, "kernel launch");
    }
    if (((uintptr_t)dev & 7u) || ((uintptr_t)host & 7u)) return fail(BPP_E_BADARG, "bpp_gather_finished: buffers must be 8-byte aligned");
    // dev == NULL: `host` is page-locked memory mapped into the device and the kernel writes the arrays there itself
    // (~200 KB of mostly consecutive stores at 65 536 bins) -- no staging buffer, no copy engine
    int groups, chunk;
    gather_shape(E, groups, chunk);
    hipLaunchKernelGGL(compact_finished_kernel, dim3(groups), dim3(kGatherThreads), 0, (hipStream_t)stream, done, ep_ret, ratio, ep_len, counter, E,
                       (unsigned char *)(dev ? dev : host), n, chunk);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "kernel launch");
    if (dev) {
        e = hipMemcpyAsync(host, dev, (size_t)BPP_FINISHED_BYTES(n), hipMemcpyDeviceToHost, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync");
    }
    e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
    if (*(const int32_t *)host != n) return fail(BPP_E_BADARG, "bpp_gather_finished: n is not the number of finished bins of this step");
    return 0;
}

int bpp_wait(void *stream) {
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? 0 : hip_fail(e, "hipStreamSynchronize");
}

// Is the page-locked allocation `p` lies in host-coherent?  (the spin-wait reads data the device wrote as soon as the
// word arrives -- correct only for coherent host memory.)  Asked once per 2 MiB-aligned address range and remembered; memory the
// runtime does not know as page-locked counts as non-coherent (the marker kernel + a final synchronisation are always right).
bool host_flag_coherent(const void *p) {
    static std::mutex m;
    static std::vector<std::pair<uintptr_t, bool>> seen;
    const uintptr_t key = (uintptr_t)p >> 21;
    {
        std::lock_guard<std::mutex> lock(m);
        for (const auto &kv : seen)
            if (kv.first == key) return kv.second;
    }
    unsigned int flags = 0;
    bool coherent = false;
    if (hipHostGetFlags(&flags, const_cast<void *>(p)) == hipSuccess) coherent = (flags & hipHostMallocNonCoherent) == 0;
    else (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(m);
    if (seen.size() > 256) seen.clear();
    seen.emplace_back(key, coherent);
    return coherent;
}

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

int bpp_mark(void *host_flag, uint32_t value, void *stream) {
    if (!host_flag || ((uintptr_t)host_flag & 3u)) return fail(BPP_E_BADARG, "bpp_mark: NULL / misaligned flag");
    // the stream memory operation is not offered by this runtime / device (hipErrorNotSupported once): marker kernel from then
    // on.  Any OTHER error of the call is the caller's (a bad stream ...) and is reported, not hidden behind the fallback.
    static std::atomic<int> use_kernel{0};
    if (!use_kernel.load(std::memory_order_relaxed) && host_flag_coherent(host_flag)) {
        const hipError_t w = hipStreamWriteValue32((hipStream_t)stream, host_flag, value, 0);
        if (w == hipSuccess) return 0;
        (void)hipGetLastError();
        if (w != hipErrorNotSupported) return hip_fail(w, "hipStreamWriteValue32");
        use_kernel.store(1, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (uint32_t *)host_flag, value);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

int bpp_wait_mark(const void *host_flag, uint32_t value, void *stream) {
    if (!host_flag) return fail(BPP_E_BADARG, "bpp_wait_mark: NULL flag");
    const uint32_t *f = (const uint32_t *)host_flag;
    const bool coherent = host_flag_coherent(host_flag);
    for (uint32_t spins = 1;; ++spins) {
        if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == value) {
            if (!coherent) {        // the data the word announces may still sit in the device's caches: let the runtime finish the job
                const hipError_t q = hipStreamSynchronize((hipStream_t)stream);
                if (q != hipSuccess) return hip_fail(q, "hipStreamSynchronize");
            }
            return 0;
        }
        if ((spins & 0xfffu) == 0) {            // every few microseconds: is the stream still alive?
            hipError_t q = hipStreamQuery((hipStream_t)stream);
            if (q == hipSuccess) {              // all done and the word not seen yet: synchronise, look once more
                q = hipStreamSynchronize((hipStream_t)stream);
                if (q != hipSuccess) return hip_fail(q, "hipStreamSynchronize");
                if (__atomic_load_n(f, __ATOMIC_ACQUIRE) == value) return 0;
                return fail(BPP_E_BADARG, "bpp_wait_mark: the stream is idle and the flag does not hold the value (no bpp_mark enqueued?)");
            }
            if (q != hipErrorNotReady) return hip_fail(q, "hipStreamQuery");
        }
        cpu_relax();
    }
}

int bpp_step_dropin(const bpp_batch *b, const int64_t *actions, const bpp_step_out *out, void *fin_host, void *host_flag,
                    uint32_t value, void *stream) {
    int rc = bpp_step(b, actions, out, stream);
    if (rc == 0 && fin_host)
        rc = bpp_gather_finished(out->done, out->ep_ret, out->ratio, out->ep_len, out->counter, b->num_envs, nullptr, fin_host,
                                 BPP_GATHER_ENQUEUE_ONLY, stream);
    if (rc == 0 && host_flag) rc = bpp_mark(host_flag, value, stream);
    return rc;
}

int bpp_rollout_uniform_sets(const bpp_batch *b, const bpp_step_out *outs, int32_t nsets, const float *first_mask,
                             int64_t *actions, uint64_t seed, uint64_t step0, int32_t nsteps, int32_t flags, void *stream) {
    if (!b || !outs || !actions || nsets < 1) return fail(BPP_E_BADARG, "bpp_rollout_uniform_sets: NULL pointer / no output set");
    if (nsteps < 0) return fail(BPP_E_BADARG, "bpp_rollout_uniform_sets: negative nsteps");
    for (int k = 0; k < nsets; ++k)
        if (!outs[k].mask) return fail(BPP_E_BADARG, "bpp_rollout_uniform_sets: every output set needs a mask");
    if (nsteps == 0) return 0;
    const int M = b->W * b->L * (1 + b->rotation);
    const uint32_t eps = BPP_ROLLOUT_EPS_OF(flags);     // SURVEY 8d's failure-path variant: one tiny launch behind every draw
    int rc = 0;
    if (!(flags & BPP_ROLLOUT_CONTINUE)) {
        if (!first_mask) return fail(BPP_E_BADARG, "bpp_rollout_uniform_sets: first_mask needed without BPP_ROLLOUT_CONTINUE");
        rc = bpp_sample_feasible(first_mask, actions, b->num_envs, M, b->env_id_base, seed, step0, stream);
        if (rc == 0 && eps) rc = bpp_epsilon_override(actions, b->num_envs, M, b->env_id_base, seed, step0, eps, stream);
    }
    for (int t = 0; rc == 0 && t < nsteps; ++t) {
        bpp_step_out o = outs[t % nsets];
        o.next_action = actions;           // every lock-step draws the next one's actions, the last one included
        o.sample_seed = seed;
        o.sample_step = step0 + (uint64_t)t + 1;
        rc = bpp_step(b, actions, &o, stream);
        if (rc == 0 && eps) rc = bpp_epsilon_override(actions, b->num_envs, M, b->env_id_base, seed, step0 + (uint64_t)t + 1, eps, stream);
    }
    return rc;
}

}  // extern "C"

namespace {
// bpp_side: what the overlapped schedule of bpp_rollout_uniform_stream needs beside the caller's stream -- ONE high-priority stream for
// the refills and three events -- created and owned by the CALLER (bpp_side_create / bpp_side_destroy; one per env), so that the
// library keeps no per-device state of its own (rounds 3-4 kept one lazily created set per device behind a mutex).
struct SideStream {
    hipStream_t stream;
    hipEvent_t stepped, refilled[2];
    int device;
};

int check_stream(const bpp_stream *s) {
    if (!s || !s->ring || !s->mt || !s->work || !s->gen_next || !s->state) return fail(BPP_E_BADARG, "bpp_stream: NULL pointer");
    if (s->num_envs <= 0 || s->depth < 4 || s->pool_len < 4 || s->env_id_base < 0)
        return fail(BPP_E_BADARG, "bpp_stream: need num_envs > 0, depth >= 4, pool_len >= 4 (two look-ahead entries, an item, the terminator)");
    if (!bpp_gen_cut2_args_ok(1, s->pool_len, s->W, s->L, s->H, s->bound_lo, s->bound_hi))
        return fail(BPP_E_BADARG, "bpp_stream: bin / bounds the reference generator cannot cut");
    if (((uintptr_t)s->ring & 3u) || ((uintptr_t)s->work & 15u) || ((uintptr_t)s->mt & 15u))
        return fail(BPP_E_BADARG, "bpp_stream: misaligned buffer");
    if (s->rng != BPP_STREAM_RNG_MT19937 && s->rng != BPP_STREAM_RNG_COUNTER) return fail(BPP_E_BADARG, "bpp_stream: unknown rng");
    return 0;
}

// The fast pipeline needs rows that hold every possible sequence (the cut kernel writes unsorted entries in place),
// rows the sort kernel can stage in LDS, and a cut-kernel workgroup that fits the LDS of a CU.
struct StreamPlan {
    bool fast;
    bool rows;              // counter generator: the rows pipeline (one lane per sequence, ranked in the lane) fits
    int maxn, cap, nsp, nslots, fb, stage;
    size_t off_jobs, off_target, off_rows, off_spill, off_twist, fast_bytes, legacy_bytes, cut_lds, sort_lds, rows_lds;
};
constexpr size_t kRowsLdsMost = 32 * 1024;   // LDS of a rows-pipeline wave; beyond it the two-kernel pipeline (cut per bin, sort) runs
StreamPlan plan_stream(const bpp_stream *s) {
    StreamPlan p{};
    const size_t E = (size_t)s->num_envs;
    p.maxn = s->W * s->L * s->H / (s->bound_lo * s->bound_lo * s->bound_lo);   // most boxes a sequence can have
    p.cap = stream_pend_cap(p.maxn);
    p.nsp = p.maxn > p.cap ? p.maxn - p.cap : 0;
    p.nslots = (int)((E + 63) / 64 + 3) * 64;
    p.off_jobs = 64;
    p.off_target = (p.off_jobs + 3 * E * 4 + 15) & ~(size_t)15;
    p.off_rows = (p.off_target + E * 4 + 15) & ~(size_t)15;
    p.off_spill = (p.off_rows + (size_t)s->depth * E * 8 + 15) & ~(size_t)15;
    p.off_twist = (p.off_spill + (size_t)2 * p.nsp * p.nslots * 4 + 15) & ~(size_t)15;
    p.fast_bytes = p.off_twist + (size_t)(p.nslots / 64) * kTwistWords * 4;     // one twist scratch per cut wave
    p.fb = stream_field_bits(s->W, s->L, s->H);
    p.stage = stream_rows_stage_cap(s->W, s->L, s->H, s->bound_lo, s->bound_hi, p.maxn);
    p.rows_lds = stream_rows_lds_bytes(p.cap, p.stage, p.fb, s->H);
    p.rows = s->rng == BPP_STREAM_RNG_COUNTER && p.rows_lds <= kRowsLdsMost &&
             (size_t)3 * (p.maxn + 2) + 2 <= stream_rows_lds_entries(p.cap, p.stage);     // a lane served alone holds any sequence
    p.legacy_bytes = (size_t)stream_work_entries(s->W, s->L, s->H, s->bound_lo) * E * 8;
    p.cut_lds = (size_t)stream_cut_lds_bytes(p.cap, p.fb);
    p.sort_lds = (size_t)4 * (s->pool_len + 256) * 4;
    p.fast = s->pool_len - 1 - kRowHdr >= p.maxn && s->pool_len <= kSortMaxT && p.cut_lds <= 64 * 1024 && p.sort_lds <= 64 * 1024;
    return p;
}
}  // namespace

extern "C" {

int bpp_stream_sizes(const bpp_stream *s, int64_t out[2]) {
    if (!s || !out) return fail(BPP_E_BADARG, "bpp_stream_sizes: NULL pointer");
    if (s->num_envs <= 0 || s->depth < 4 || !bpp_gen_cut2_args_ok(1, s->pool_len, s->W, s->L, s->H, s->bound_lo, s->bound_hi))
        return fail(BPP_E_BADARG, "bpp_stream_sizes: fill in num_envs, depth, pool_len, the bin and the bounds first");
    if (s->rng != BPP_STREAM_RNG_MT19937 && s->rng != BPP_STREAM_RNG_COUNTER) return fail(BPP_E_BADARG, "bpp_stream_sizes: unknown rng");
    const StreamPlan p = plan_stream(s);
    out[0] = (int64_t)(s->rng == BPP_STREAM_RNG_COUNTER ? kCtrRec : kMtRec) * s->num_envs;
    out[1] = (int64_t)(p.fast_bytes > p.legacy_bytes ? p.fast_bytes : p.legacy_bytes);
    return 0;
}

int bpp_stream_init(const bpp_stream *s, void *stream) {
    int rc = check_stream(s);
    if (rc) return rc;
    hipLaunchKernelGGL(stream_init_kernel, dim3((s->num_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, *s);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

}  // extern "C"

namespace {
// kmax / urgent: see StreamWork.  The plain kernel always brings every bin to `depth` rows.
int stream_refill(const bpp_stream *s, void *stream, int kmax, int urgent) {
    int rc = check_stream(s);
    if (rc) return rc;
    const StreamPlan p = plan_stream(s);
    hipStream_t st = (hipStream_t)stream;
    if (!p.fast || current_knobs().stream_legacy == 1) {
        hipLaunchKernelGGL(stream_refill_kernel, dim3((s->num_envs + kStreamLanes - 1) / kStreamLanes), dim3(kStreamLanes),
                           (size_t)kStreamLdsWords * kStreamLanes * 4, st, *s);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
    }
    unsigned char *base = (unsigned char *)s->work;
    const StreamWork w0{(int32_t *)base, (int32_t *)(base + p.off_jobs), (int32_t *)(base + p.off_target), (int64_t *)(base + p.off_rows),
                       (uint32_t *)(base + p.off_spill), (uint32_t *)(base + p.off_twist), p.cap, p.nsp, p.nslots, p.maxn, p.fb, kmax, urgent, p.stage};
    hipError_t e = hipMemsetAsync(base, 0, 64, st);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync");
    const int E = s->num_envs;
    hipLaunchKernelGGL(stream_scan_kernel, dim3((E + kScanThreads - 1) / kScanThreads), dim3(kScanThreads), 2 * (kScanThreads / 64) * 4 * sizeof(int), st,
                       *s, w0);
    if (p.rows && current_knobs().stream_legacy != 2) {     // counter generator, one lane per sequence, no sort kernel
        StreamWork w = w0;
        if (current_knobs().stream_legacy == 3) w.cap = w.cap < 10 ? w.cap : 10, w.stage = w.stage < 24 ? w.stage : 24;
        const int64_t waves = ((int64_t)E * s->depth + 63) / 64;
        // one wave per 64 rows, all at once: a grid bounded to 256 / 512 / 1 024 waves that walk through the rows -- the refill
        // spread over more of the window -- costs the lock-steps beside it MORE (profiles/r5x_*: -10 / -6 / -2 %)
        const unsigned grid = (unsigned)(waves < 8192 ? waves : 8192);
        if (p.fb == 4) hipLaunchKernelGGL(stream_cut_rows_kernel<4>, dim3(grid), dim3(64), p.rows_lds, st, *s, w);
        else hipLaunchKernelGGL(stream_cut_rows_kernel<8>, dim3(grid), dim3(64), p.rows_lds, st, *s, w);
        e = hipGetLastError();
        return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
    }
    const StreamWork &w = w0;
    if (s->rng == BPP_STREAM_RNG_COUNTER) {      // no generator state: nothing to regenerate, the lists are all the LDS a cut wave needs
        const size_t lds = (size_t)stream_cut_lds_bytes(p.cap, p.fb) - (size_t)64 * kRingStride;
        if (p.fb == 4) hipLaunchKernelGGL(stream_cut_ctr_kernel<4>, dim3(p.nslots / 64), dim3(64), lds, st, *s, w);
        else hipLaunchKernelGGL(stream_cut_ctr_kernel<8>, dim3(p.nslots / 64), dim3(64), lds, st, *s, w);
    } else {
        hipLaunchKernelGGL(stream_pretwist_kernel, dim3((unsigned)((E + 3) / 4 < 2048 ? (E + 3) / 4 : 2048)), dim3(256),
                           4 * kTwistWords * sizeof(uint32_t), st, *s, w);
        if (p.fb == 4) hipLaunchKernelGGL(stream_cut_kernel<4>, dim3(p.nslots / 64), dim3(64), p.cut_lds, st, *s, w);
        else hipLaunchKernelGGL(stream_cut_kernel<8>, dim3(p.nslots / 64), dim3(64), p.cut_lds, st, *s, w);
    }
    const int64_t most = ((int64_t)E * s->depth + 3) / 4;
    hipLaunchKernelGGL(stream_sort_kernel, dim3((unsigned)(most < 2048 ? most : 2048)), dim3(256), p.sort_lds, st, *s, w);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}
}  // namespace

extern "C" {

int bpp_stream_refill(const bpp_stream *s, void *stream) { return stream_refill(s, stream, 0, 0); }

int bpp_side_create(void **side) {
    if (!side) return fail(BPP_E_BADARG, "bpp_side_create: NULL pointer");
    *side = nullptr;
    SideStream *ss = new SideStream();
    if (hipGetDevice(&ss->device) != hipSuccess) {
        delete ss;
        return fail(BPP_E_BADARG, "bpp_side_create: no current device");
    }
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    // (a CU mask on this stream -- refills confined to 32 .. 128 CUs so that the others keep all their workgroup slots
    // for the step kernel -- was measured: 0.33 - 0.74 G env steps/s against 1.17 - 1.28 G without, the refill becomes
    // the critical path)
    hipError_t e = hipStreamCreateWithPriority(&ss->stream, hipStreamNonBlocking, greatest);
    if (e != hipSuccess) {
        delete ss;
        return hip_fail(e, "hipStreamCreateWithPriority");
    }
    hipEvent_t *ev[3] = {&ss->stepped, &ss->refilled[0], &ss->refilled[1]};
    for (int k = 0; k < 3; ++k) {
        e = hipEventCreateWithFlags(ev[k], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int j = 0; j < k; ++j) (void)hipEventDestroy(*ev[j]);
            (void)hipStreamDestroy(ss->stream);
            delete ss;
            return hip_fail(e, "hipEventCreateWithFlags");
        }
    }
    *side = ss;
    return 0;
}

int bpp_side_destroy(void *side) {
    if (!side) return 0;
    SideStream *ss = (SideStream *)side;
    (void)hipStreamSynchronize(ss->stream);
    (void)hipEventDestroy(ss->stepped);
    (void)hipEventDestroy(ss->refilled[0]);
    (void)hipEventDestroy(ss->refilled[1]);
    hipError_t e = hipStreamDestroy(ss->stream);
    delete ss;
    return e == hipSuccess ? 0 : hip_fail(e, "hipStreamDestroy");
}

int bpp_rollout_uniform_stream(const bpp_batch *b, const bpp_step_out *out, int64_t *actions, uint64_t seed, uint64_t step0,
                               int32_t nsteps, const bpp_stream *s, int32_t refill_every, void *side_handle, void *stream) {
    if (!b || !s || !out || !out->mask || !actions) return fail(BPP_E_BADARG, "bpp_rollout_uniform_stream: NULL pointer");
    if (nsteps < 0) return fail(BPP_E_BADARG, "bpp_rollout_uniform_stream: negative nsteps");
    const int behind = b->seq_cache ? 4 : 3;   // rows a step launch may touch from the current one on (a cache line refers to the row after next)
    if (b->pool_mode != BPP_POOL_RING || refill_every < 1 || refill_every > s->depth - behind)
        return fail(BPP_E_BADARG, "bpp_rollout_uniform_stream: needs a ring pool and 1 <= refill_every <= depth - 3 (- 4 with seq_cache)");
    int rc = 0;
    // (3 below stands for `behind`.)  With depth >= 2 R + 3 rows per bin the refill that follows a chunk of R lock-steps may run BESIDE the next chunk
    // (it only rewrites rows of finished episodes): it goes to a side stream, and a chunk starts once the refill issued two
    // chunks earlier is complete.  Margin m = rows a bin has from its current episode on when a refill scans it (the
    // previous refill is complete by then: same stream).  A bin advances by at most R episodes per chunk and a step reads
    // two rows ahead, so it needs m >= R + 3 to get through the chunk that runs beside the refill and m + need >= 2 R + 3
    // to get through the one after (this refill complete, the next one running).  The scan guarantees the second
    // (need >= 2 R + 3 - m, `urgent`), which also gives the first for the next scan: m' >= m + need - R >= R + 3.
    SideStream *side = (current_knobs().stream_overlap && s->depth >= 2 * refill_every + behind) ? (SideStream *)side_handle : nullptr;
    hipStream_t main = (hipStream_t)stream;
    if (side) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != side->device)
            return fail(BPP_E_BADARG, "bpp_rollout_uniform_stream: the bpp_side was created on another device");
    }
    int32_t chunk = 0;
    for (int32_t done = 0; rc == 0 && done < nsteps; done += refill_every, ++chunk) {
        const int32_t n = nsteps - done < refill_every ? nsteps - done : refill_every;
        if (side && chunk >= 2) (void)hipStreamWaitEvent(main, side->refilled[chunk & 1], 0);
        // only the first chunk draws its first action with a launch of its own: the last step of every chunk but the last draws
        // the next chunk's (the refill in between touches no mask) -- a sampler launch reads the whole mask, 8 / 50 us for 10x10 / 20x20
        rc = rollout_uniform_steps(b, out, actions, seed, step0 + (uint64_t)done, n, chunk == 0, done + n < nsteps, stream);
        if (rc) break;
        if (!side) {
            rc = bpp_stream_refill(s, stream);
            continue;
        }
        (void)hipEventRecord(side->stepped, main);
        (void)hipStreamWaitEvent(side->stream, side->stepped, 0);
        // beside the lock-steps a short refill matters more than a full ring: a bin gets about twice what the average
        // bin uses in refill_every lock-steps (one sequence per ~9), more only if it would otherwise run out before the
        // refill after the next one is complete
        rc = stream_refill(s, side->stream, refill_every < 7 ? 2 : (refill_every + 5) / 6, 2 * refill_every + behind);
        (void)hipEventRecord(side->refilled[chunk & 1], side->stream);
    }
    if (side) {     // everything enqueued on `stream` after this call sees the refilled ring
        if (chunk >= 2) (void)hipStreamWaitEvent(main, side->refilled[chunk & 1], 0);
        if (chunk >= 1) (void)hipStreamWaitEvent(main, side->refilled[(chunk - 1) & 1], 0);
        hipError_t e = hipGetLastError();
        if (rc == 0 && e != hipSuccess) rc = hip_fail(e, "side-stream refill");
        if (rc == 0) rc = bpp_stream_refill(s, stream);     // leave every bin with `depth` rows, as the serial schedule does
    }
    return rc;
}

int bpp_episode_stats(const uint8_t *done, const double *ep_ret, const double *ratio, const int32_t *ep_len, int32_t E,
                      double *acc, void *stream) {
    if (!done || !ep_ret || !ratio || !ep_len || !acc) return fail(BPP_E_BADARG, "bpp_episode_stats: NULL pointer");
    if (E <= 0) return fail(BPP_E_BADARG, "bpp_episode_stats: non-positive size");
    hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(BPP_REDUCE_LANES), 0, (hipStream_t)stream, done, ep_ret, ratio, ep_len, E, acc);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}

int bpp_episode_acc_reduce(double *ep_acc, int32_t E, double *acc, int32_t clear, void *scratch, void *stream) {
    if (!ep_acc || !acc) return fail(BPP_E_BADARG, "bpp_episode_acc_reduce: NULL pointer");
    if (E <= 0) return fail(BPP_E_BADARG, "bpp_episode_acc_reduce: non-positive size");
    if ((uintptr_t)ep_acc & 31u) return fail(BPP_E_BADARG, "bpp_episode_acc_reduce: ep_acc must be 32-byte aligned");
    if (scratch != nullptr) {
        if ((uintptr_t)scratch & 7u) return fail(BPP_E_BADARG, "bpp_episode_acc_reduce: scratch must be 8-byte aligned");
        hipLaunchKernelGGL(acc_reduce_wide_kernel, dim3(kAccWideGroups), dim3(256), 0, (hipStream_t)stream, ep_acc, E, acc, clear,
                           (double *)scratch);
    } else {
        hipLaunchKernelGGL(acc_reduce_kernel, dim3(1), dim3(BPP_REDUCE_LANES), 0, (hipStream_t)stream, ep_acc, E, acc, clear);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "kernel launch");
}


}  // extern "C"
