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

template <bool VEC, int MODE>
__global__ __launch_bounds__(kWave * kWavesPerBlock) void bpp_kernel(const Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x >> 6;
    const int e0 = (xcd_block(p.xcd_remap) * (blockDim.x >> 6) + wid) * p.epw;
    if (e0 >= p.E) return;  // no block-level barrier is ever used, a whole wave may leave
    const int nenv = min(p.epw, p.E - e0);
    const int A = p.A, L = p.L, M = p.M;
    unsigned char *wb = smem + wid * p.lds_per_wave;
    uint8_t *hm = wb;                        // [epw][A] heights
    uint8_t *mk = wb + p.off_mk;             // [epw][M] feasibility bytes
    BinRec *rec = (BinRec *)(wb + p.off_rec);  // [epw]
    const int ncell = nenv * A;
    constexpr int GW = VEC ? 4 : 1;          // cells handled per lane per access

    // ---- phase 1: stage this wave's heightmaps into LDS as bytes -------------------------------
    if (MODE == kStep) {
        const uint8_t *gh = p.hmap + (size_t)e0 * A;
        if (VEC) {
            for (int q = lane; q < ncell / 4; q += kWave) ((uint32_t *)hm)[q] = ((const uint32_t *)gh)[q];
        } else {
            for (int c = lane; c < ncell; c += kWave) hm[c] = gh[c];
        }
    } else if (MODE == kMaskHmap) {
        const int32_t *gh = p.hmap_in + (size_t)e0 * A;
        if (VEC) {
            for (int q = lane; q < ncell / 4; q += kWave) {
                int4 v = ((const int4 *)gh)[q];
                ((uint32_t *)hm)[q] = min((uint32_t)v.x, 255u) | (min((uint32_t)v.y, 255u) << 8) | (min((uint32_t)v.z, 255u) << 16) |
                                      (min((uint32_t)v.w, 255u) << 24);
            }
        } else {
            for (int c = lane; c < ncell; c += kWave) hm[c] = (uint8_t)min((uint32_t)gh[c], 255u);
        }
    } else if (MODE == kMaskObs) {
        // acktr/utils.py:41-47: plane 0 of the observation row is the heightmap
        if (VEC) {
            for (int q = lane; q < ncell / 4; q += kWave) {
                uint32_t el = p.divA4.div(q);  // bin within the wave (A/4 quads per bin)
                float4 v = ((const float4 *)(p.obs_in + (size_t)(e0 + el) * 4 * A))[q - el * (A / 4)];
                ((uint32_t *)hm)[q] = min((uint32_t)(int)v.x, 255u) | (min((uint32_t)(int)v.y, 255u) << 8) |
                                      (min((uint32_t)(int)v.z, 255u) << 16) | (min((uint32_t)(int)v.w, 255u) << 24);
            }
        } else {
            for (int c = lane; c < ncell; c += kWave) {
                uint32_t el = p.divA.div(c);
                hm[c] = (uint8_t)min((uint32_t)(int)p.obs_in[(size_t)(e0 + el) * 4 * A + (c - el * A)], 255u);
            }
        }
    } else {
        if (VEC) {
            for (int q = lane; q < ncell / 4; q += kWave) ((uint32_t *)hm)[q] = 0u;  // space.py:22
        } else {
            for (int c = lane; c < ncell; c += kWave) hm[c] = 0;
        }
    }
    wave_sync();

    // ---- phase 2: lane-per-bin scalar work ------------------------------------------------------
    bool fin = false;
    double fin_ret = 0.0, fin_ratio = 0.0;
    int fin_len = 0;
    if (lane < nenv) {
        const int e = e0 + lane;
        BinRec r;
        r.place = 0;
        r.flags = 0;
        r.any = 0;
        if (MODE == kStep) {
            bpp_env_state st = p.state[e];
            const int64_t act = p.actions[e];
            // BoxCreator.preview(1)[0] (binCreator.py:15-18): the current item, the one after it and the
            // first item of the next episode are cached in the state record; the entries the NEXT step
            // will need are fetched here, speculatively for both outcomes, off the critical path.
            const int T = p.T;
            int seq_n = st.seq + p.seq_stride;
            seq_n = seq_n >= p.P ? seq_n - p.P : seq_n;
            int seq_nn = seq_n + p.seq_stride;
            seq_nn = seq_nn >= p.P ? seq_nn - p.P : seq_nn;
            const uint32_t it_cur = st.item_cur, it_nxt = st.item_next, it_rst = st.item_reset;
            const LookAheadAt la = look_ahead_at(p, st.seq, seq_n, seq_nn, st.cursor);
            const uint32_t sp_ok = p.pool[la.ok], sp_f1 = p.pool[la.f1], sp_f2 = p.pool[la.f2];
            const int ix = it_cur & 255, iy = (it_cur >> 8) & 255, iz = (it_cur >> 16) & 255;
            // bin3D.py:96-105: rotated iff idx > area (strict)
            const bool noop = act == BPP_ACTION_NOOP;   // include/bpp_abi.h: the bin is left alone
            int64_t idx = act;
            const bool flag = p.rotation && idx > A;
            if (flag) idx -= A;
            const int x = flag ? iy : ix, y = flag ? ix : iy, z = iz;  // space.py:166-172
            bool ok = idx >= 0 && idx < (int64_t)(p.W + 1) * L;
            int lx = 0, ly = 0, top = 0;
            if (ok) {
                lx = (int)p.divL.div((uint32_t)idx);  // space.py:153-156
                ly = (int)idx - lx * L;
                ok = (lx + x <= p.W) && (ly + y <= L);  // space.py:112-115
            }
            if (ok) {
                Win w = scan_window(hm + lane * A, L, lx, ly, x, y);
                ok = feasible(w, x * y, z, p.H, BPP_RULE_SPACE);  // space.py:117-144
                top = w.mh + z;                                   // space.py:42-45 with lz = max_h
            }
            const int vol = ix * iy * iz;
            // bin3D.py:44-46,108-121: float64 (vol / binvol) * 10, 0.0 on failure
            const double rew = ok ? ((double)vol / p.binvol) * 10.0 : 0.0;
            st.n_boxes += ok ? 1 : 0;
            st.vol_sum += ok ? vol : 0;
            st.ep_ret = st.ep_ret + rew;  // bench/monitor.py:58-62 (sum in step order)
            st.ep_len += noop ? 0 : 1;
            p.reward[e] = (float)rew;     // acktr/envs.py:192
            p.done[e] = (ok || noop) ? 0 : 1;
            if (p.host_reward) {
                p.host_reward[e] = (float)rew;
                p.host_done[e] = (ok || noop) ? 0 : 1;
            }
            p.counter[e] = st.n_boxes;    // bin3D.py:111,124
            p.ratio[e] = (double)st.vol_sum / p.binvol;  // space.py:146-151
            p.ep_ret[e] = st.ep_ret;
            p.ep_len[e] = st.ep_len;
            fin = !ok && !noop;
            fin_ret = st.ep_ret;
            fin_ratio = (double)st.vol_sum / p.binvol;
            fin_len = st.ep_len;
            if (ok) {
                st.cursor += 1;  // bin3D.py:116-117
                st.item_cur = it_nxt;
                st.item_next = sp_ok;
                st.hmax = max(st.hmax, (uint32_t)top);   // highest cell of the bin
                r.item = it_nxt;
                r.place = (uint32_t)lx | ((uint32_t)ly << 8) | ((uint32_t)x << 16) | ((uint32_t)y << 24);
                r.flags = 1u | ((uint32_t)top << 8);
            } else if (noop) {
                r.item = it_cur;
            } else {  // shmem_vec_env.py:128-129 auto-reset; bin3D.py:55-59
                st.episode += 1;
                st.seq = seq_n;
                st.cursor = 0;
                st.n_boxes = 0;
                st.vol_sum = 0;
                st.ep_ret = 0.0;
                st.ep_len = 0;
                st.item_cur = it_rst;
                st.item_next = sp_f1;
                st.item_reset = sp_f2;
                st.hmax = 0;
                r.item = it_rst;
                r.flags = 2u;
            }
            p.state[e] = st;
            if (p.cache != nullptr) row_cache_drop(p, e);
        } else if (MODE == kResetInit || MODE == kResetAdvance) {
            bpp_env_state st;
            if (MODE == kResetInit) {
                st.episode = 0;
                st.seq = (int32_t)(((uint32_t)p.base_mod + (uint32_t)e) % (uint32_t)p.P);
            } else {
                st = p.state[e];
                st.episode += 1;
                int s = st.seq + p.seq_stride;
                st.seq = s >= p.P ? s - p.P : s;
            }
            st.cursor = 0;
            st.n_boxes = 0;
            st.vol_sum = 0;
            st.ep_ret = 0.0;
            st.ep_len = 0;
            int sn = st.seq + p.seq_stride;
            sn = sn >= p.P ? sn - p.P : sn;
            st.item_cur = p.pool[(size_t)st.seq * p.T + p.ring2];
            st.item_next = p.pool[(size_t)st.seq * p.T + p.ring2 + min(1, p.T - 1 - p.ring2)];
            st.item_reset = p.pool[(size_t)sn * p.T + p.ring2];
            st.hmax = 0;
            p.state[e] = st;
            if (p.cache != nullptr) row_cache_drop(p, e);
            r.item = st.item_cur;
            r.flags = 2u;
        } else if (MODE == kMaskObs) {
            // acktr/utils.py:43-45: x, y, z = int(plane[k][0])
            const float *o = p.obs_in + (size_t)e * 4 * A;
            r.item = pack_item((int)o[A], (int)o[2 * A], (int)o[3 * A]);
        } else {
            const int32_t *it = p.items_in + (size_t)e * 3;
            r.item = pack_item(it[0], it[1], it[2]);
        }
        rec[lane] = r;
    }
    if (MODE == kStep && p.ep_acc && fin) episode_acc_add(p.ep_acc, e0 + lane, fin_ret, fin_ratio, fin_len);
    wave_sync();

    if (MODE == kStep || MODE == kResetInit || MODE == kResetAdvance) {
        // ---- phase 3a: apply the placement / reset to the LDS tile (space.py:36-46) -------------
        if (MODE == kStep) {
            for (int g = lane; g < ncell / GW; g += kWave) {
                uint32_t packed = VEC ? ((uint32_t *)hm)[g] : (uint32_t)hm[g];
                uint32_t outv = 0;
#pragma unroll
                for (int k = 0; k < GW; ++k) {
                    const uint32_t c = g * GW + k;
                    const uint32_t el = p.divA.div(c);
                    const uint32_t cell = c - el * A;
                    const uint32_t i = p.divL.div(cell), j = cell - i * L;
                    const BinRec r = rec[el];
                    uint32_t v = (packed >> (8 * k)) & 255u;
                    const uint32_t lx = r.place & 255u, ly = (r.place >> 8) & 255u;
                    const uint32_t x = (r.place >> 16) & 255u, y = r.place >> 24;
                    if ((r.flags & 1u) && (i - lx) < x && (j - ly) < y) v = r.flags >> 8;
                    if (r.flags & 2u) v = 0;
                    outv |= v << (8 * k);
                }
                if (VEC) ((uint32_t *)hm)[g] = outv;
                else hm[g] = (uint8_t)outv;
            }
            wave_sync();
        }
        // ---- phase 3b: stream out the byte heightmap (state) and the float32 observation -----------
        // bin3D.py:49-66: planes [hmap, x, y, z]; float32 at the VecEnv buffer (shmem_vec_env.py:42-43)
        {
            uint8_t *gh = p.hmap + (size_t)e0 * A;
            float *go = p.obs + (size_t)e0 * 4 * A;
            const int per_plane = A / GW;
            for (int g = lane; g < nenv * 4 * per_plane; g += kWave) {
                const uint32_t pl = p.divA4.div(g);  // plane counter: bin*4 + plane
                const uint32_t k = g - pl * per_plane;
                const uint32_t el = pl >> 2, plane = pl & 3u;
                if (plane == 0) {
                    if (VEC) {
                        const uint32_t v = ((uint32_t *)hm)[el * per_plane + k];
                        ((uint32_t *)gh)[el * per_plane + k] = v;
                        ((float4 *)go)[g] = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u),
                                                        (float)(v >> 24));
                    } else {
                        const int v = hm[el * A + k];
                        gh[el * A + k] = (uint8_t)v;
                        go[g] = (float)v;
                    }
                } else {
                    const float f = (float)((rec[el].item >> (8 * (plane - 1))) & 255u);
                    if (VEC) ((float4 *)go)[g] = make_float4(f, f, f, f);
                    else go[g] = f;
                }
            }
        }
        if (p.mask == nullptr) return;
    }

    // ---- phase 4: feasibility of every candidate position (acktr/utils.py:37-94) ---------------
    for (int c = lane; c < nenv * M; c += kWave) {
        const uint32_t el = p.divM.div(c);
        uint32_t r = c - el * M;
        const bool rot = r >= (uint32_t)A;  // second half: item turned by 90 degrees, utils.py:81-89
        if (rot) r -= A;
        const uint32_t i = p.divL.div(r), j = r - i * L;
        const uint32_t item = rec[el].item;
        const int ix = item & 255u, iy = (item >> 8) & 255u, z = (item >> 16) & 255u;
        const int x = rot ? iy : ix, y = rot ? ix : iy;
        bool f = false;
        if (x >= 1 && y >= 1 && (int)i + x <= p.W && (int)j + y <= L) {  // utils.py:54-55 loop ranges (a zero-sized side never fits, like the fast path)
            Win w = scan_window(hm + el * A, L, i, j, x, y);
            f = feasible(w, x * y, z, p.H, p.rule);
        }
        mk[c] = f ? 1 : 0;
        if (f) rec[el].any = 1u;
    }
    wave_sync();

    // ---- phase 5: float32 mask out, all-ones when nothing is feasible (utils.py:59-60,91-92) ---
    {
        float *gm = p.mask + (size_t)e0 * M;
        if (VEC) {
            const int per = M / 4;
            for (int g = lane; g < nenv * per; g += kWave) {
                const uint32_t el = p.divA4.div(p.rotation ? (g >> 1) : g);  // g / (M/4)
                const uint32_t v = rec[el].any ? ((uint32_t *)mk)[g] : 0x01010101u;
                ((float4 *)gm)[g] = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u),
                                                (float)(v >> 24));
            }
        } else {
            for (int c = lane; c < nenv * M; c += kWave) {
                const uint32_t el = p.divM.div(c);
                gm[c] = rec[el].any ? (float)mk[c] : 1.0f;
            }
        }
    }
}


// =================================================================================================
// Fast path: compile-time geometry + packed-histogram integral image
// =================================================================================================
// The generic kernel above walks every candidate's x*y window cell by cell (14 instructions and one
// LDS round trip per cell).  Here every cell of height h is coded as the 64-bit integer 1 << (5*h)
// -- a histogram over height levels with 5-bit counters -- and a 2-D inclusive prefix sum P of the
// codes is built in LDS once per step (two scans).  The histogram of ANY window of <= 31 cells is then
//   P[i+x][j+y] - P[i][j+y] - P[i+x][j] + P[i][j]          (plain 64-bit integer arithmetic; prefix
// totals may overflow a field, the final difference cannot), its top set field is max_h and that
// field's value is max_area: 4 LDS reads + 3 subtractions + one clz per candidate, no loop.
// 12 levels fit one word (H <= 10 leaves level H+1 for out-of-range inputs), K words cover
// H + 2 <= 12*K.  Windows of more than 31 cells (the bin-sized terminator item) are tiled into
// <= 5x6 pieces whose (max, count) pairs are merged.
constexpr int kFieldBits = 5;
constexpr int kLevelsPerWord = 12;
constexpr int kTileX = 5, kTileY = 6;

template <int K>
struct __attribute__((aligned(8 * K))) Ent {
    uint64_t w[K];
};

template <int K>
__device__ __forceinline__ Ent<K> code_of(uint32_t h) {
    Ent<K> c;
    if (K == 1) {
        c.w[0] = 1ull << (kFieldBits * h);
    } else {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const uint32_t rel = h - kLevelsPerWord * k;  // wraps to a huge value when h is below this word
            c.w[k] = rel < (uint32_t)kLevelsPerWord ? 1ull << (kFieldBits * rel) : 0ull;
        }
    }
    return c;
}

// One-word codes of a bin whose heights need two words, one word at a time (bpp_tile_kernel's two-phase scan of tall
// 20x20 bins): PH = 1 -> the upper word (levels 12..23, zero for lower cells), PH = 2 -> the lower word (levels 0..11,
// zero for higher cells); PH = 0 -> the plain one-word code (every height <= 11).
template <int PH>
__device__ __forceinline__ Ent<1> code_phase(uint32_t h) {
    Ent<1> c;
    if (PH == 0) {
        c.w[0] = 1ull << (kFieldBits * h);
    } else {
        const uint32_t rel = PH == 1 ? h - (uint32_t)kLevelsPerWord : h;   // wraps to a huge value below the upper word
        c.w[0] = rel < (uint32_t)kLevelsPerWord ? 1ull << (kFieldBits * rel) : 0ull;
    }
    return c;
}

// Highest non-empty level of a window histogram and the count stored there.
template <int K>
__device__ __forceinline__ void top_of(const Ent<K> &h, int &m, int &cnt) {
    uint64_t v = h.w[0];
    int word = 0;
#pragma unroll
    for (int k = 1; k < K; ++k)
        if (h.w[k] != 0) {
            v = h.w[k];
            word = k;
        }
    const int msb = 63 - __builtin_clzll(v);
    const int lvl = (msb * 13) >> 6;  // msb / 5 for msb <= 63
    cnt = (int)(v >> (kFieldBits * lvl));  // lvl is the top non-empty field: nothing above it to mask off
    m = word * kLevelsPerWord + lvl;
}

template <int K, bool ZERO_OK = false>
__device__ __forceinline__ void rect_top(const Ent<K> *P00, int PW, int xa, int yb, int &m, int &cnt) {
    const Ent<K> a = P00[0], b = P00[yb], c = P00[xa * PW], d = P00[xa * PW + yb];
    Ent<K> h;
#pragma unroll
    for (int k = 0; k < K; ++k) h.w[k] = d.w[k] - b.w[k] - c.w[k] + a.w[k];
    top_of<K>(h, m, cnt);
    if (ZERO_OK && K == 1 && h.w[0] == 0ull) {   // no cell of this rectangle has a level in the word scanned: contributes nothing
        m = -1;
        cnt = 0;
    }
}

// (max_h, max_area) of window [i,i+x) x [j,j+y) from the bin's prefix image (acktr/utils.py:14-16).
template <int K, bool ZERO_OK = false>
__device__ __forceinline__ void window_top(const Ent<K> *Pb, int PW, int i, int j, int x, int y, int &mh, int &ma) {
    if (x <= kTileX && y <= kTileY) {
        rect_top<K, ZERO_OK>(Pb + i * PW + j, PW, x, y, mh, ma);
        return;
    }
    mh = -1;
    ma = 0;
    // (rare path -- items wider than 5 x 6, in the benchmark only the bin-sized terminator: kept rolled, an unrolled
    // copy of these loops was what set the kernels' scalar-register count and cost the 10x10 + rotation kernel a workgroup slot per CU)
#pragma unroll 1
    for (int a0 = 0; a0 < x; a0 += kTileX) {
        const int xa = min(kTileX, x - a0);
#pragma unroll 1
        for (int b0 = 0; b0 < y; b0 += kTileY) {
            const int yb = min(kTileY, y - b0);
            int m, c;
            rect_top<K, ZERO_OK>(Pb + (i + a0) * PW + (j + b0), PW, xa, yb, m, c);
            ma = m > mh ? c : ma + (m == mh ? c : 0);
            mh = max(mh, m);
        }
    }
}

// Prefix image of ONE bin by a whole wave (used when a wave owns a single bin, e.g. the 20x20 bin
// whose image is 7 KB): every row is split into SEG = 64/W segments so that W*SEG (60 of 64 for W = 20)
// lanes scan concurrently; each lane scans its <= CS cells, the segment totals are exchanged with
// __shfl_up and added as offsets.  Same for the column pass.  (With lane-per-row scans only W of 64 lanes
// worked and this phase was 27 % of the 20^3 step.)
template <int K>
__device__ __forceinline__ Ent<K> shfl_up_ent(const Ent<K> &v, int d) {
    Ent<K> r;
#pragma unroll
    for (int k = 0; k < K; ++k) r.w[k] = (uint64_t)__shfl_up((unsigned long long)v.w[k], d, kWave);
    return r;
}

template <int W, int L, int K, int PH = 0>
__device__ __forceinline__ void build_prefix_one_bin(const uint8_t *hm, Ent<K> *P, uint32_t hclamp, int lane) {
    constexpr int PW = L + 1;
    constexpr int SR = (kWave / W) < 1 ? 1 : (kWave / W), CSR = (L + SR - 1) / SR;  // row pass: segments along j
    constexpr int SC = (kWave / L) < 1 ? 1 : (kWave / L), CSC = (W + SC - 1) / SC;  // column pass: segments along i
    Ent<K> zero;
#pragma unroll
    for (int k = 0; k < K; ++k) zero.w[k] = 0;
    for (int t = lane; t < PW + W; t += kWave) P[t < PW ? t : (t - PW + 1) * PW] = zero;  // row 0, column 0
    {
        const int i = lane / SR, sg = lane - i * SR;
        const bool act = i < W;
        const int j0 = sg * CSR;
        const uint8_t *row = hm + (act ? i : 0) * L;
        Ent<K> s[CSR];
        Ent<K> run = zero;
#pragma unroll
        for (int c = 0; c < CSR; ++c) {
            const int j = j0 + c;
            if (j < L) {
                Ent<K> cd;
                if constexpr (K == 1 && PH != 0) cd = code_phase<PH>(min((uint32_t)row[j], hclamp));
                else cd = code_of<K>(min((uint32_t)row[j], hclamp));
#pragma unroll
                for (int k = 0; k < K; ++k) run.w[k] += cd.w[k];
            }
            s[c] = run;
        }
        Ent<K> off = zero;
#pragma unroll
        for (int d = 1; d < SR; ++d) {
            const Ent<K> t = shfl_up_ent<K>(run, d);
            if (sg >= d) {
#pragma unroll
                for (int k = 0; k < K; ++k) off.w[k] += t.w[k];
            }
        }
        if (act) {
            Ent<K> *pr = P + (i + 1) * PW + 1;
#pragma unroll
            for (int c = 0; c < CSR; ++c)
                if (j0 + c < L) {
                    Ent<K> o;
#pragma unroll
                    for (int k = 0; k < K; ++k) o.w[k] = s[c].w[k] + off.w[k];
                    pr[j0 + c] = o;
                }
        }
    }
    wave_sync();
    {
        const int j = lane / SC, sg = lane - j * SC;
        const bool act = j < L;
        const int i0 = sg * CSC;
        Ent<K> *pc = P + PW + ((act ? j : 0) + 1);
        Ent<K> s[CSC];
        Ent<K> run = zero;
#pragma unroll
        for (int c = 0; c < CSC; ++c) {
            const int i = i0 + c;
            if (i < W) {
                const Ent<K> v = pc[i * PW];
#pragma unroll
                for (int k = 0; k < K; ++k) run.w[k] += v.w[k];
            }
            s[c] = run;
        }
        Ent<K> off = zero;
#pragma unroll
        for (int d = 1; d < SC; ++d) {
            const Ent<K> t = shfl_up_ent<K>(run, d);
            if (sg >= d) {
#pragma unroll
                for (int k = 0; k < K; ++k) off.w[k] += t.w[k];
            }
        }
        if (act) {
#pragma unroll
            for (int c = 0; c < CSC; ++c)
                if (i0 + c < W) {
                    Ent<K> o;
#pragma unroll
                    for (int k = 0; k < K; ++k) o.w[k] = s[c].w[k] + off.w[k];
                    pc[(i0 + c) * PW] = o;
                }
        }
    }
    wave_sync();
}

// Per-bin, per-orientation constants of the item shown in the next observation, computed once by the
// bin's lane and read (one ds_read_b128) by every candidate lane.  The float64 ratio tests of
// acktr/utils.py:28-33 become integer thresholds on max_area:
//   ma/area > 0.95  <=>  ma >= floor(19*area/20) + 1   (t95), likewise t85 (17/20) and t50 (1/2).
constexpr int kCandShift = 22;  // candidate index decode, see make_ori
struct __attribute__((aligned(16))) OriRec {
    uint32_t a;  // x | y<<8 | (max(H - z + 1, 0))<<16 (9 bits) | big<<25 | valid<<26
    uint32_t b;  // t95 | t85<<16
    uint32_t c;  // t50 | (W - x)<<16 | (L - y)<<24
    uint32_t d;
};

__device__ __forceinline__ OriRec make_ori(int W, int L, int x, int y, int z, int H) {
    OriRec o;
    const int area = x * y;
    const uint32_t valid = (x >= 1 && y >= 1 && x <= W && y <= L) ? 1u : 0u;
    const uint32_t big = (x > kTileX || y > kTileY) ? 1u : 0u;
    const uint32_t hz1 = (uint32_t)max(H - z + 1, 0);
    o.a = (uint32_t)x | ((uint32_t)y << 8) | (hz1 << 16) | (big << 25) | (valid << 26);
    o.b = (uint32_t)(19 * area / 20 + 1) | ((uint32_t)(17 * area / 20 + 1) << 16);
    o.c = (uint32_t)(area / 2 + 1) | ((uint32_t)((W - x) & 255) << 16) | ((uint32_t)((L - y) & 255) << 24);
    // ceil(2^22 / nj): t / nj == (t * d) >> 22 for every t < 1024, nj <= 256 (t * d < 2^32, d < 2^24: one
    // full-rate 24-bit multiply; exhaustively checked in tests/test_host_logic.py)
    o.d = ((1u << kCandShift) + (uint32_t)(L - y + 1) - 1u) / (uint32_t)max(L - y + 1, 1);
    return o;
}

template <int W, int L, int K, bool ROT, int MODE>
__global__ __launch_bounds__(kWave * kMaxFastWavesPerBlock) void bpp_fast_kernel(const Params p) {
    // W == 0 selects the runtime-geometry instantiation: sizes come from the launch parameters and the
    // divisions below use their precomputed magic numbers; with W, L > 0 everything folds to constants.
    constexpr bool RT = (W == 0);
    static_assert(RT || (W * L) % 4 == 0, "fast path needs W*L % 4 == 0");
    const int Wv = RT ? p.W : W, Lv = RT ? p.L : L;
    const int A = Wv * Lv, A4 = A / 4, M = ROT ? 2 * A : A, M4 = M / 4, PW = Lv + 1, PN = (Wv + 1) * (Lv + 1);
    auto div_a4 = [&](int n) { return RT ? (int)p.divA4.div((uint32_t)n) : n / A4; };
    auto div_m4 = [&](int n) { return RT ? (int)p.divM4.div((uint32_t)n) : n / M4; };
    auto div_l = [&](int n) { return RT ? (int)p.divL.div((uint32_t)n) : n / Lv; };
    auto div_w = [&](int n) { return RT ? (int)p.divW.div((uint32_t)n) : n / Wv; };
    auto div_pww = [&](int n) { return RT ? (int)p.divPWW.div((uint32_t)n) : n / (PW + Wv); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int blk_e0 = xcd_block(p.xcd_remap) * wpb * p.epw;       // first bin of this workgroup
    const int e0 = blk_e0 + wid * p.epw;                           // first bin of this wave
    const int nenv = max(0, min(p.epw, p.E - e0));                 // block barriers below: no early return
    unsigned char *wb = smem + wid * p.lds_per_wave;
    uint8_t *hm = wb;
    uint32_t *hm32 = (uint32_t *)wb;
    uint8_t *mk = wb + p.off_mk;
    BinRec *rec = (BinRec *)(wb + p.off_rec);
    OriRec *ori = (OriRec *)(wb + p.off_ori);  // [epw][2]
    Ent<K> *P = (Ent<K> *)(wb + p.off_P);
    const uint32_t hclamp = (uint32_t)p.H + 1u;  // heights above H all behave like H+1 (never feasible)

    if (BPP_ABL(p, 16)) return;
    // The deciding wave (wave 0) issues its per-bin loads first, so their latency overlaps the staging.
    const int dec_nb = max(0, min(wpb * p.epw, p.E - blk_e0));
    const int dec_e = blk_e0 + (lane < dec_nb ? lane : 0);
    bpp_env_state st0;
    int64_t act0 = 0;
    if (MODE == kStep && wid == 0 && !BPP_ABL(p, 32)) {
        st0 = p.state[dec_e];
        act0 = p.actions[dec_e];
    }

    // ---- phase 1: stage heightmaps as bytes ------------------------------------------------------
    if (MODE == kStep) {
        const uint32_t *gh = (const uint32_t *)(p.hmap + (size_t)e0 * A);
        for (int q = lane; q < (BPP_ABL(p, 64) ? 0 : nenv * A4); q += kWave) hm32[q] = gh[q];
    } else if (MODE == kMaskHmap) {
        const int4 *gh = (const int4 *)(p.hmap_in + (size_t)e0 * A);
        for (int q = lane; q < nenv * A4; q += kWave) {
            const int4 v = gh[q];
            hm32[q] = min((uint32_t)v.x, 255u) | (min((uint32_t)v.y, 255u) << 8) | (min((uint32_t)v.z, 255u) << 16) |
                      (min((uint32_t)v.w, 255u) << 24);
        }
    } else if (MODE == kMaskObs) {
        for (int q = lane; q < nenv * A4; q += kWave) {
            const int el = div_a4(q);
            const float4 v = ((const float4 *)(p.obs_in + (size_t)(e0 + el) * 4 * A))[q - el * A4];
            hm32[q] = min((uint32_t)(int)v.x, 255u) | (min((uint32_t)(int)v.y, 255u) << 8) |
                      (min((uint32_t)(int)v.z, 255u) << 16) | (min((uint32_t)(int)v.w, 255u) << 24);
        }
    } else {
        for (int q = lane; q < nenv * A4; q += kWave) hm32[q] = 0u;  // space.py:22
    }
    __syncthreads();  // wave 0 reads the other waves' tiles below

    // ---- phase 2: per-bin scalar work, lane-per-bin, done by ONE wave for the whole workgroup ------
    // A workgroup owns wpb * epw (<= 64) consecutive bins.  Wave 0 carries one bin per lane through the
    // scalar chain (state, action, items, placement rule, reward, Monitor, next item) and leaves a
    // record per bin in the owning wave's LDS area; the other waves wait at the barrier.  (Executing
    // this chain redundantly in every wave, for only `epw` bins each, cost ~4x the VALU work of this phase.)
    bool fin = false;
    double fin_ret = 0.0, fin_ratio = 0.0;
    int fin_len = 0;
    if (wid == 0 && !BPP_ABL(p, 32)) {
        const bool active = lane < dec_nb;
        const int e = dec_e;
        const int ow = lane >> p.epw_shift, oel = lane & (p.epw - 1);  // owning wave, bin within it
        unsigned char *ob = smem + ow * p.lds_per_wave;
        const uint8_t *ohm = ob + oel * A;
        BinRec r;
        r.item = 0;
        r.place = 0;
        r.flags = 0;
        r.any = 0;
        if (MODE == kStep) {
            bpp_env_state st = st0;                                    // loaded before the tile was staged
            const int64_t act = act0;
            // binCreator.py:15-18: current / next / first-of-next-episode items come from the state record;
            // the pool entries the NEXT step needs are fetched speculatively for both outcomes.
            const int T = p.T;
            int seq_n = st.seq + p.seq_stride;
            seq_n = seq_n >= p.P ? seq_n - p.P : seq_n;
            int seq_nn = seq_n + p.seq_stride;
            seq_nn = seq_nn >= p.P ? seq_nn - p.P : seq_nn;
            const uint32_t it_cur = st.item_cur, it_nxt = st.item_next, it_rst = st.item_reset;
            const LookAheadAt la = look_ahead_at(p, st.seq, seq_n, seq_nn, st.cursor);
            const uint32_t sp_ok = p.pool[la.ok], sp_f1 = p.pool[la.f1], sp_f2 = p.pool[la.f2];
            const int ix = it_cur & 255, iy = (it_cur >> 8) & 255, iz = (it_cur >> 16) & 255;
            const bool noop = act == BPP_ACTION_NOOP;                  // include/bpp_abi.h: the bin is left alone
            int64_t idx = act;                                         // bin3D.py:96-105
            const bool flag = ROT && idx > A;
            if (flag) idx -= A;
            const int x = flag ? iy : ix, y = flag ? ix : iy, z = iz;  // space.py:166-172
            bool ok = active && idx >= 0 && idx < (int64_t)(Wv + 1) * Lv;
            int lx = 0, ly = 0;
            if (ok) {
                lx = div_l((int)idx);                                  // space.py:153-156
                ly = (int)idx - lx * Lv;
                ok = (lx + x <= Wv) && (ly + y <= Lv);                 // space.py:112-115
            }
            int top = 0;
            if (ok) {
                const uint8_t *hb = ohm + lx * Lv + ly;
                int mh = 0, ma = 0;                                    // space.py:127-129
                if (x <= 5 && y <= 5) {
                    // common item sizes: 25 predicated independent LDS reads instead of a divergent loop
                    int v[5][5];
#pragma unroll
                    for (int a = 0; a < 5; ++a)
#pragma unroll
                        for (int b = 0; b < 5; ++b) v[a][b] = (a < x && b < y) ? (int)hb[a * Lv + b] : -1;
#pragma unroll
                    for (int a = 0; a < 5; ++a)
#pragma unroll
                        for (int b = 0; b < 5; ++b) mh = max(mh, v[a][b]);
#pragma unroll
                    for (int a = 0; a < 5; ++a)
#pragma unroll
                        for (int b = 0; b < 5; ++b) ma += (v[a][b] == mh);
                } else {
                    for (int a = 0; a < x; ++a)
                        for (int b = 0; b < y; ++b) {
                            const int v = hb[a * Lv + b];
                            ma = v > mh ? 1 : ma + (v == mh);
                            mh = max(mh, v);
                        }
                }
                const int r00 = hb[0], r10 = hb[(x - 1) * Lv], r01 = hb[y - 1], r11 = hb[(x - 1) * Lv + y - 1];
                const int rm = max(max(r00, r10), max(r01, r11));      // space.py:117-125
                Win w;
                w.mh = mh;
                w.ma = ma;
                w.c = (r00 == mh) + (r10 == mh) + (r01 == mh) + (r11 == mh);
                w.sc = (r00 == rm) + (r10 == rm) + (r01 == rm) + (r11 == rm);
                ok = feasible(w, x * y, z, p.H, BPP_RULE_SPACE);       // space.py:131-144
                top = mh + z;                                          // space.py:42-45 with lz = max_h
            }
            const int vol = ix * iy * iz;
            const double rew = ok ? ((double)vol / p.binvol) * 10.0 : 0.0;  // bin3D.py:44-46,108-121
            st.n_boxes += ok ? 1 : 0;
            st.vol_sum += ok ? vol : 0;
            st.ep_ret = st.ep_ret + rew;                               // bench/monitor.py:58-62
            st.ep_len += noop ? 0 : 1;
            const double ratio = (double)st.vol_sum / p.binvol;        // space.py:146-151
            if (active) {
                p.reward[e] = (float)rew;                              // acktr/envs.py:192
                p.done[e] = (ok || noop) ? 0 : 1;
                if (p.host_reward) {
                    p.host_reward[e] = (float)rew;
                    p.host_done[e] = (ok || noop) ? 0 : 1;
                }
                p.counter[e] = st.n_boxes;                             // bin3D.py:111,124
                p.ratio[e] = ratio;
                p.ep_ret[e] = st.ep_ret;
                p.ep_len[e] = st.ep_len;
            }
            fin = active && !ok && !noop;
            fin_ret = st.ep_ret;
            fin_ratio = ratio;
            fin_len = st.ep_len;
            if (ok) {
                st.cursor += 1;                                        // bin3D.py:116-117
                st.item_cur = it_nxt;
                st.item_next = sp_ok;
                st.hmax = max(st.hmax, (uint32_t)top);                 // highest cell of the bin
                r.item = it_nxt;
                r.place = (uint32_t)lx | ((uint32_t)ly << 8) | ((uint32_t)x << 16) | ((uint32_t)y << 24);
                r.flags = 1u | ((uint32_t)top << 8);
            } else if (noop) {
                r.item = it_cur;
            } else {                                                   // shmem_vec_env.py:128-129
                st.episode += 1;
                st.seq = seq_n;
                st.cursor = 0;
                st.n_boxes = 0;
                st.vol_sum = 0;
                st.ep_ret = 0.0;
                st.ep_len = 0;
                st.item_cur = it_rst;
                st.item_next = sp_f1;
                st.item_reset = sp_f2;
                st.hmax = 0;
                r.item = it_rst;
                r.flags = 2u;
            }
            if (active) p.state[e] = st;
            if (active && p.cache != nullptr) row_cache_drop(p, e);
        } else if (MODE == kResetInit || MODE == kResetAdvance) {
            bpp_env_state st;
            if (MODE == kResetInit) {
                st.episode = 0;
                st.seq = (int32_t)(((uint32_t)p.base_mod + (uint32_t)e) % (uint32_t)p.P);
            } else {
                st = p.state[e];
                st.episode += 1;
                const int sq = st.seq + p.seq_stride;
                st.seq = sq >= p.P ? sq - p.P : sq;
            }
            st.cursor = 0;
            st.n_boxes = 0;
            st.vol_sum = 0;
            st.ep_ret = 0.0;
            st.ep_len = 0;
            int sn = st.seq + p.seq_stride;
            sn = sn >= p.P ? sn - p.P : sn;
            st.item_cur = p.pool[(size_t)st.seq * p.T + p.ring2];
            st.item_next = p.pool[(size_t)st.seq * p.T + p.ring2 + min(1, p.T - 1 - p.ring2)];
            st.item_reset = p.pool[(size_t)sn * p.T + p.ring2];
            st.hmax = 0;
            if (active) p.state[e] = st;
            if (active && p.cache != nullptr) row_cache_drop(p, e);
            r.item = st.item_cur;
            r.flags = 2u;
        } else if (MODE == kMaskObs) {
            const float *o = p.obs_in + (size_t)e * 4 * A;             // acktr/utils.py:43-45
            r.item = pack_item((int)o[A], (int)o[2 * A], (int)o[3 * A]);
        } else {
            const int32_t *it = p.items_in + (size_t)e * 3;
            r.item = pack_item(it[0], it[1], it[2]);
        }
        if (active) {
            ((BinRec *)(ob + p.off_rec))[oel] = r;
            OriRec *oo = (OriRec *)(ob + p.off_ori) + oel * 2;
            const int nx = r.item & 255u, ny = (r.item >> 8) & 255u, nz = (r.item >> 16) & 255u;
            oo[0] = make_ori(Wv, Lv, nx, ny, nz, p.H);
            if (ROT) oo[1] = make_ori(Wv, Lv, ny, nx, nz, p.H);          // utils.py:81-84
        }
    }
    __syncthreads();
    // episode statistics (main.py:159-162): off the other waves' critical path, after the barrier
    if (MODE == kStep && wid == 0 && p.ep_acc && fin && !BPP_ABL(p, 128))
        episode_acc_add(p.ep_acc, dec_e, fin_ret, fin_ratio, fin_len);

    if (MODE == kStep) {
        // ---- phase 2b: every wave applies its bins' placements (space.py:36-46: window := max_h + z),
        // one sub-group of 64/epw lanes per bin, rows split over the sub-group ---------------------
        const int G = kWave >> p.epw_shift;
        const int el = lane >> (6 - p.epw_shift), sl = lane & (G - 1);
        if (el < nenv) {
            const BinRec r = rec[el];
            if (r.flags & 1u) {
                const int lx = r.place & 255u, ly = (r.place >> 8) & 255u, x = (r.place >> 16) & 255u, y = r.place >> 24;
                uint8_t *hb = hm + el * A + lx * Lv + ly;
                const uint8_t top = (uint8_t)(r.flags >> 8);
                for (int a = sl; a < x; a += G)
                    for (int b = 0; b < y; ++b) hb[a * Lv + b] = top;
            }
        }
        wave_sync();
    }

    auto write_obs = [&]() {
        uint32_t *gh = (uint32_t *)(p.hmap + (size_t)e0 * A);
        float4 *go = (float4 *)(p.obs + (size_t)e0 * 4 * A);
        for (int q = lane; q < nenv * A4; q += kWave) {
            const int el = div_a4(q);
            const uint32_t v = hm32[q];
            gh[q] = v;
            go[q + el * (3 * A4)] = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u),
                                                (float)(v >> 24));
        }
        // planes x, y, z are constants per bin (bin3D.py:49-53): bin-uniform passes, the value comes from a
        // scalar register and every lane keeps one fixed offset
        for (int el = 0; el < nenv; ++el) {
            const uint32_t item = __builtin_amdgcn_readfirstlane(rec[el].item);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const float f = (float)((item >> (8 * pl)) & 255u);
                const float4 v = make_float4(f, f, f, f);
                float4 *gp = go + el * A + (pl + 1) * A4;
                for (int k = lane; k < A4; k += kWave) gp[k] = v;
            }
        }
    };

    if (MODE == kStep || MODE == kResetInit || MODE == kResetAdvance) {
        if (MODE == kStep) {
            // ---- phase 3a: finished bins restart from an empty map --------------------------------
            for (int q = lane; q < nenv * A4; q += kWave)
                if (rec[div_a4(q)].flags & 2u) hm32[q] = 0u;
            wave_sync();
        }
        // ---- phase 3b: byte heightmap (state) + float32 observation out (bin3D.py:49-66) ----------
        if (!BPP_ABL(p, 8)) write_obs();
        if (p.mask == nullptr) return;
    }

    // ---- phase 4a: prefix image of the height-level codes ------------------------------------------
    bool built = false;
    if constexpr (!RT) {
        if (!BPP_ABL(p, 1) && p.epw == 1 && W * 2 <= kWave) {
            if (nenv > 0) build_prefix_one_bin<W, L, K>(hm, P, hclamp, lane);
            built = true;
        }
    }
    if (!built && !BPP_ABL(p, 1)) {
        Ent<K> zero;
#pragma unroll
        for (int k = 0; k < K; ++k) zero.w[k] = 0;
        for (int t = lane; t < nenv * (PW + Wv); t += kWave) {         // row 0 and column 0
            const int el = div_pww(t), r = t - el * (PW + Wv);
            P[el * PN + (r < PW ? r : (r - PW + 1) * PW)] = zero;
        }
        for (int t = lane; t < nenv * Wv; t += kWave) {                // running sums along each row
            const int el = div_w(t), i = t - el * Wv;
            const uint8_t *row = hm + el * A + i * Lv;
            Ent<K> *pr = P + el * PN + (i + 1) * PW + 1;
            Ent<K> s = zero;
            if constexpr (!RT) {
                uint32_t hv[L > 0 ? L : 1];
#pragma unroll
                for (int j = 0; j < L; ++j) hv[j] = row[j];
#pragma unroll
                for (int j = 0; j < L; ++j) {
                    const Ent<K> c = code_of<K>(min(hv[j], hclamp));
#pragma unroll
                    for (int k = 0; k < K; ++k) s.w[k] += c.w[k];
                    pr[j] = s;
                }
            } else {
                for (int j = 0; j < Lv; ++j) {
                    const Ent<K> c = code_of<K>(min((uint32_t)row[j], hclamp));
#pragma unroll
                    for (int k = 0; k < K; ++k) s.w[k] += c.w[k];
                    pr[j] = s;
                }
            }
        }
        wave_sync();
        for (int t = lane; t < nenv * Lv; t += kWave) {                // then down each column
            const int el = div_l(t), j = t - el * Lv;
            Ent<K> *pc = P + el * PN + PW + (j + 1);
            Ent<K> s = zero;
            if constexpr (!RT) {
                constexpr int CH = W % 10 == 0 ? 10 : (W % 5 == 0 ? 5 : 1);
                for (int i0 = 0; i0 < W; i0 += CH) {
                    Ent<K> v[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) v[i] = pc[(i0 + i) * PW];
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
#pragma unroll
                        for (int k = 0; k < K; ++k) s.w[k] += v[i].w[k];
                        pc[(i0 + i) * PW] = s;
                    }
                }
            } else {
                for (int i = 0; i < Wv; ++i) {
                    const Ent<K> v = pc[i * PW];
#pragma unroll
                    for (int k = 0; k < K; ++k) s.w[k] += v.w[k];
                    pc[i * PW] = s;
                }
            }
        }
        wave_sync();
    }

    // ---- phase 4b: feasibility of every candidate position (acktr/utils.py:37-94) ----------------
    // Bin-uniform evaluation: the wave walks its bins (and orientations) one after the other, so the
    // item constants live in scalar registers, and only the (W-x+1)*(L-y+1) in-range candidates
    // (utils.py:54-55 loop ranges) are enumerated -- lane t <-> (i, j) = (t / nj, t % nj).
    for (int g = lane; g < nenv * M4; g += kWave) ((uint32_t *)mk)[g] = 0u;
    wave_sync();
    for (int el = 0; el < (BPP_ABL(p, 2) ? 0 : nenv); ++el) {
        unsigned long long any = 0ull;
        const Ent<K> *Pe = P + el * PN;
        const uint8_t *he = hm + el * A;
        uint8_t *me = mk + el * M;
        // a bin that was just reset shows an empty map: its mask is the in-range rectangle (no lookups)
        const bool fresh = (MODE == kStep || MODE == kResetInit || MODE == kResetAdvance) &&
                           (__builtin_amdgcn_readfirstlane(rec[el].flags) & 2u) != 0u;
#pragma unroll
        for (int rot = 0; rot < (ROT ? 2 : 1); ++rot) {                // utils.py:81-89: second half
            const OriRec ov = ori[el * 2 + rot];
            const uint32_t oa = __builtin_amdgcn_readfirstlane(ov.a), ob = __builtin_amdgcn_readfirstlane(ov.b),
                           oc = __builtin_amdgcn_readfirstlane(ov.c), od = __builtin_amdgcn_readfirstlane(ov.d);
            if (!(oa & (1u << 26))) continue;                          // item does not fit at all
            const int x = oa & 255u, y = (oa >> 8) & 255u, hz1 = (oa >> 16) & 511u;
            if (ROT && rot == 1 && x == y) {
                // square footprint: the turned item's mask (utils.py:81-89) equals the first half
                for (int g = lane; g < A4; g += kWave) ((uint32_t *)me)[A4 + g] = ((const uint32_t *)me)[g];
                continue;
            }
            const bool big = (oa >> 25) & 1u;
            const int nj = (int)(oc >> 24) + 1, nv = ((int)((oc >> 16) & 255u) + 1) * nj;
            const int t95 = ob & 0xffffu, t85 = ob >> 16, t50 = oc & 0xffffu;
            const int o10 = (x - 1) * Lv, o01 = y - 1;
            // one candidate loop per case, so that no bin-uniform condition is re-tested per candidate
            auto run = [&](auto big_c, auto empty_c) {
                constexpr bool BIG = decltype(big_c)::value, EMPTY = decltype(empty_c)::value;
#pragma unroll 2
                for (int t = lane; t < nv; t += kWave) {
                    const int i = (int)(((uint32_t)t * od) >> kCandShift), j = t - i * nj;
                    bool f;
                    if (EMPTY) {
                        f = hz1 > 0;  // empty map: max_h = 0 over the whole window, every in-range position passes
                    } else {
                        const Ent<K> *Pb = Pe + i * PW + j;
                        int mh, ma;
                        if (!BIG) {
                            const Ent<K> a = Pb[0], b = Pb[y], cc = Pb[x * PW], d = Pb[x * PW + y];
                            Ent<K> h;
#pragma unroll
                            for (int k = 0; k < K; ++k) h.w[k] = (a.w[k] + d.w[k]) - (b.w[k] + cc.w[k]);
                            top_of<K>(h, mh, ma);
                        } else {
                            window_top<K>(Pe, PW, i, j, x, y, mh, ma);
                        }
                        const uint8_t *hb = he + i * Lv + j;
                        const int r00 = hb[0], r10 = hb[o10], r01 = hb[o01], r11 = hb[o10 + o01];
                        const int cnt = (r00 == mh) + (r10 == mh) + (r01 == mh) + (r11 == mh);  // utils.py:23-26
                        const int thr = cnt == 4 ? t50 : (cnt == 3 ? t85 : t95);
                        f = (mh < hz1) && (ma >= thr);                 // utils.py:20-33
                        if (p.rule == BPP_RULE_SPACE) {                // space.py:122-125: sc >= 3
                            const int rm = max(max(r00, r10), max(r01, r11));
                            f = f && ((r00 == rm) + (r10 == rm) + (r01 == rm) + (r11 == rm) >= 3);
                        }
                    }
                    me[rot * A + i * Lv + j] = f ? 1 : 0;
                    any |= __ballot(f);
                }
            };
            using T = std::true_type;
            using F = std::false_type;
            if (fresh)
                run(F{}, T{});
            else if (big)
                run(T{}, F{});
            else
                run(F{}, F{});
        }
        if (any != 0ull && lane == 0) rec[el].any = 1u;
    }
    wave_sync();

    // ---- phase 4c (optional): draw the next action uniformly among the feasible entries ------------
    // Same result as bpp_sample_feasible on the mask this step writes: one sub-group of 64/epw lanes per
    // bin; a lane owns `per` consecutive dwords (4 mask bytes each) of the bin's LDS mask, byte sums come
    // from one multiply (bytes are 0/1), an inclusive scan inside the sub-group locates the lane holding
    // the pick-th set entry (pick = hash * count >> 32) and prefix-byte compares locate it in the dword.
    if (MODE == kStep && p.next_action != nullptr) {
        const int NQ = M4;
        const int G = kWave >> p.epw_shift;
        const int el = lane >> (6 - p.epw_shift), sl = lane & (G - 1);
        const bool act = el < nenv;
        const int per = (NQ + G - 1) / G;
        const uint32_t *mq = (const uint32_t *)mk + (act ? el : 0) * NQ;
        const bool anyf = act && rec[act ? el : 0].any != 0u;
        int cnt = 0;
        for (int k = 0; k < per; ++k) {
            const int q = sl * per + k;
            const uint32_t v = (act && q < NQ) ? (anyf ? mq[q] : 0x01010101u) : 0u;
            cnt += (int)((v * 0x01010101u) >> 24);
        }
        int incl = cnt;
        for (int d = 1; d < G; d <<= 1) {
            const int o = __shfl_up(incl, d, kWave);
            if (sl >= d) incl += o;
        }
        const int total = __shfl(incl, lane | (G - 1), kWave);
        const int e = e0 + el;
        int rem = (int)__umulhi(mix32(mix32_base(p.sample_seed, p.sample_step), (uint32_t)(p.env_id_base + e)), (uint32_t)total) -
                  (incl - cnt);
        if (act && total > 0 && rem >= 0 && rem < cnt) {
            int found = 0;
            for (int k = 0; k < per; ++k) {
                const int q = sl * per + k;
                const uint32_t v = q < NQ ? (anyf ? mq[q] : 0x01010101u) : 0u;
                const uint32_t cum = v * 0x01010101u;          // byte t = number of set entries among bytes 0..t
                const int c = (int)(cum >> 24);
                if (rem >= 0 && rem < c) {
                    // first byte whose running count exceeds rem
                    const uint32_t r = (uint32_t)rem;
                    found = q * 4 + (int)(((cum & 255u) <= r) + (((cum >> 8) & 255u) <= r) + (((cum >> 16) & 255u) <= r));
                }
                rem -= c;
            }
            p.next_action[e] = found;
        }
    }

    // ---- phase 5: float32 mask out, all-ones fallback (utils.py:59-60,91-92) ---------------------
    {
        float4 *gm = (float4 *)(p.mask + (size_t)e0 * M);
        for (int g = lane; g < (BPP_ABL(p, 4) ? 0 : nenv * M4); g += kWave) {
            const uint32_t v = rec[div_m4(g)].any ? ((const uint32_t *)mk)[g] : 0x01010101u;
            gm[g] = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24));
        }
    }
}

#include "bpp_tile_kernel.inl"
#include "bpp_stream_gen.inl"

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
// Round 6 (profiles/r6e_sweep_bins_by_tile_groups_*.txt, two boxes): with ONE group per wave a launch of 262 144 / 1 048 576 10x10 bins
// costs 14 - 17 % more per bin than a 65 536-bin launch (146 - 152 us instead of 4 x 31.8; eight and more rounds of workgroups: the
// later rounds' cold reads queue behind the earlier rounds' write streams), with two or four groups per wave it does not (128 - 132 /
// 124 - 132 us; 1 048 576 bins: 477 - 494 / 466 - 475 us = 2.1 - 2.25 G env steps/s) -- fewer, longer workgroups whose loads all
// go out before any of their stores.  With rotation two groups win (155 vs 162 us with four, 175 with one).
constexpr TileGeoEntry kTileGeo[] = {{10, 10, 1, 4, 1, 4, 2}, {20, 20, 1, 1, 1, 4, 4}, {20, 20, 2, 1, 1, 4, 4}, {10, 10, 2, 4, 1, 2, 2}};
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
