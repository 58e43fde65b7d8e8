// bpp_tile_kernel.inl -- the production step/reset/mask kernel for the compile-time geometries (10x10 and 20x20
// bins), included by bpp_kernels.hip inside its anonymous namespace.  Same algorithm as bpp_fast_kernel (byte
// tiles in LDS, one deciding wave per workgroup, packed-histogram prefix image, bin-uniform candidate loop) with
// the instruction count cut where the round-1 counters said it went (the kernel is VALU-issue bound):
//   * everything that depends only on the lane is a compile-time constant: G = 64 / EPW lanes per bin, a lane
//     owns quads sl, sl + G, sl + 2G ... of its bin, so every tile / observation / mask access is ONE per-lane
//     base address plus immediate offsets, and the plane a store belongs to is known per unrolled iteration
//     (round 1: 12 stores at 25 of 64 lanes for the constant planes, an index division per access);
//   * the deciding wave spreads the placement window over LPB = 64 / (bins per workgroup) lanes per bin and
//     merges (max, count) with two shuffles per halving instead of 25 predicated reads in one lane;
//   * per-orientation constants (thresholds, candidate ranges, the index-decode multiplier from a table in
//     constant memory) are derived on the scalar unit from the bin's item, not in vector registers;
//   * the corner-count rule is evaluated on lane masks (scalar and/or), two selects per candidate;
//   * the uniform-feasible draw of the next action works on the candidate loop's ballots (scalar popcounts,
//     one mbcnt in the winning pass) instead of re-scanning the LDS mask bytes;
//   * episode statistics: the (few) finishing lanes add their four values to the bin's own accumulator row;
//   * a wave owns NIT groups of EPW bins and walks them one after the other: all tiles are staged up front (one
//     latency for all), wave 0 decides ALL bins of the workgroup in one pass behind ONE pair of barriers (so the
//     other waves idle once per 4*EPW*NIT bins, not once per 4*EPW), and the stores of group i overlap the
//     compute of group i+1.  Wave 0 issues its global stores only after the second barrier: the speculative pool
//     loads they depend on must not hold the other waves up.
// Reference semantics are cited at the same places as in bpp_fast_kernel.

// ceil(2^22 / n) for n = 1..256 (n = 0 unused): the candidate index decode i = (t * magic) >> 22, see make_ori.
struct CandMagicTable {
    uint32_t v[257];
};
constexpr CandMagicTable make_cand_magic() {
    CandMagicTable t{};
    t.v[0] = 0;
    for (uint32_t n = 1; n <= 256; ++n) t.v[n] = ((1u << kCandShift) + n - 1u) / n;
    return t;
}
__constant__ const CandMagicTable kCandMagic = make_cand_magic();

constexpr int kTileWaves = 4;  // waves per workgroup of the tile kernel
// Where a finishing bin's episode-accumulator row is read and written back (A/B of round 3, profiles/archive/r3x_ab_r3aa.txt;
// the rejected forms live there, not here).  kAccLate (a wave owns several bins): every bin reads its row up front with
// the state record AS A PREFETCH (the values are dead unless the bin finishes), the finishing bins read it again behind
// the second barrier -- now a cache hit -- and add / store it at the END of the kernel: 10x10 past the Infinity Cache
// 33.1 -> 30.6 us.  One bin per wave (20x20, already at its memory floor there): read behind the second barrier, added
// and stored at once.
// Constants of one (bin, orientation) of the item on display: computed lane-parallel for all the wave's bins at once
// (lane sl == orientation of bin el holds the slot's seven words in registers) and read back wave-uniformly by the
// candidate loop with v_readlane -- no LDS round trip, ~100 scalar instructions per bin saved.
//   w0: index-decode multiplier ceil(2^22 / nj) (23 bits)
//   w1: nv (candidates, 11 bits) | nj << 11 (5 bits) | valid << 28 | big << 29 | (x == y) << 31
//   w2: x * PW entries (prefix-image row offset) | y << 16
//   w3: (x - 1) * L (corner offset) | (y - 1) << 16
//   w4: t95 | t85 << 16          w5: t50 | x << 16 | y << 24
//   w6 (per step): hz1 << 16 (9 bits: max(H - z + 1, 0)) | fresh << 30 | (second half of a rotation mask) << 31
//   w7: hash32(seed, global bin id, step) of the fused draw
// Per-bin record in LDS written by the deciding wave (or, in the mask-only modes, by the owning wave), read by the
// bin's lanes: 12 bytes.
struct TileRec {
    uint32_t item;   // item shown in the next observation: x | y<<8 | z<<16
    uint32_t place;  // lx | ly<<8 | x<<16 | y<<24 of the box just placed
    uint32_t flags;  // bit0 placed, bit1 reset (zero the map), bit2 every height of the bin <= kLowTop, bits 8.. new top height
};
// A bin none of whose heights exceeds 11 fits ONE 64-bit histogram word (12 levels of 5 bits): the 20x20x20 kernel
// (K = 2) then builds and scans a K = 1 prefix image for it -- 83 % of the (bin, lock-step) pairs under the
// benchmark's policy (episodes end long before the pallet is half full), half the prefix-image work and LDS traffic.
constexpr int kLowTop = kLevelsPerWord - 1;

constexpr int round16(int v) { return (v + 15) & ~15; }

// The six words of a slot that the candidate loop consumes, from the item on display (same code for the
// lane-parallel and the scalar-unit evaluation).  Everything that depends on the footprint (x, y) alone -- the candidate
// ranges, the index-decode multiplier, the offsets, the three integer thresholds of SURVEY.md A.3 -- comes from a table in
// constant memory, one 32-byte entry per footprint, built at compile time for the two tile geometries (round 4 computed
// them per launch and bin: ~60 of a wave's ~680 vector instructions, executed by 4 of its 64 lanes); what is left per slot
// is the table index, the height bound hz1 and three flag bits.
struct alignas(32) SlotEntry {   // one entry = one 32-byte line; make_slot_words reads it as a 16-byte and an 8-byte load
    uint32_t w[8];   // w0 .. w5 as documented above with hz1 = 0 and fresh = 0; bit 31 of w1 = "x == y" (square candidate); w6, w7 unused
};
template <int W, int L>
struct SlotTable {
    SlotEntry e[(W + 1) * (L + 1)];   // entry x * (L + 1) + y; row 0 and column 0 (and entry 0 for every footprint that does not fit): "invalid"
};
template <int W, int L>
constexpr SlotTable<W, L> make_slot_table() {
    constexpr int PW = L + 1;
    SlotTable<W, L> t{};
    for (int x = 0; x <= W; ++x)
        for (int y = 0; y <= L; ++y) {
            SlotEntry &s = t.e[x * (L + 1) + y];
            const bool valid = x >= 1 && y >= 1;
            const uint32_t nj = valid ? (uint32_t)(L - y + 1) : 1u, nv = valid ? (uint32_t)(W - x + 1) * nj : 0u;   // utils.py:54-55 loop ranges
            const uint32_t area = valid ? (uint32_t)(x * y) : 0u;
            // floor(k * area / 20) + 1
            const uint32_t t95 = 19u * area / 20u + 1u, t85 = 17u * area / 20u + 1u, t50 = (area >> 1) + 1u;
            const bool big = x > kTileX || y > kTileY;
            s.w[0] = ((1u << kCandShift) + nj - 1u) / nj;
            s.w[1] = nv | (nj << 11) | ((uint32_t)valid << 28) | ((uint32_t)(valid && big) << 29) | ((uint32_t)(valid && x == y) << 31);
            s.w[2] = (uint32_t)(x * PW) | ((uint32_t)y << 16);
            s.w[3] = (uint32_t)(x > 0 ? (x - 1) * L : 0) | ((uint32_t)(y > 0 ? y - 1 : 0) << 16);
            s.w[4] = t95 | (t85 << 16);
            s.w[5] = t50 | ((uint32_t)x << 16) | ((uint32_t)y << 24);
            s.w[6] = 0u;
            s.w[7] = 0u;
        }
    return t;
}
__constant__ const SlotTable<10, 10> kSlotTable10 = make_slot_table<10, 10>();
__constant__ const SlotTable<20, 20> kSlotTable20 = make_slot_table<20, 20>();

template <int W, int L>
__device__ __forceinline__ void make_slot_words(uint32_t item, int rot, bool fresh, bool rot_kernel, int H, uint32_t w[7]) {
    static_assert((W == 10 && L == 10) || (W == 20 && L == 20), "a slot table exists for the tile geometries only");
    const uint32_t ix = item & 255u, iy = (item >> 8) & 255u, z = (item >> 16) & 255u;
    const uint32_t x = rot ? iy : ix, y = rot ? ix : iy;
    const bool fits = x <= (uint32_t)W && y <= (uint32_t)L;                   // (x == 0 or y == 0: an "invalid" entry of the table)
    const uint32_t at = fits ? x * (uint32_t)(L + 1) + y : 0u;
    const SlotEntry *ent;
    if constexpr (W == 10) ent = &kSlotTable10.e[at];
    else ent = &kSlotTable20.e[at];
    // w0 .. w5 are the entry AS LOADED -- nothing is computed from them here, so that the loads' latency is only waited for
    // where the candidate loop reads the words (behind the observation store and the prefix image), not in this phase;
    // what depends on the step (z, a fresh bin, which half of a rotation mask) travels in a word of its own
    const uint4 lo = *(const uint4 *)&ent->w[0];
    const uint2 hi = *(const uint2 *)&ent->w[4];
    w[0] = lo.x, w[1] = lo.y, w[2] = lo.z, w[3] = lo.w, w[4] = hi.x, w[5] = hi.y;
    w[6] = ((uint32_t)max(H - (int)z + 1, 0) << 16) | ((uint32_t)fresh << 30) | ((uint32_t)(rot_kernel && rot == 1) << 31);
}

template <int W, int L, int K, bool ROT, int EPW, int NIT>
struct TileGeo {
    static constexpr int A = W * L, A4 = A / 4, M = ROT ? 2 * A : A, M4 = M / 4, PW = L + 1, PN = (W + 1) * (L + 1);
    static constexpr int G = kWave / EPW;              // lanes per bin in a tile wave
    static constexpr int NBW = EPW * NIT;              // bins per wave: NIT groups of EPW bins, one after the other
    static constexpr int NB = kTileWaves * NBW;        // bins per workgroup
    static constexpr int LPB = kWave / NB;             // lanes per bin in the deciding wave
    static constexpr int NPASS = (A + kWave - 1) / kWave;  // candidate passes per orientation (at most A candidates)
    static constexpr int OFF_MK = round16(NBW * A);                              // after the NBW byte tiles
    static constexpr int OFF_REC = round16(OFF_MK + EPW * M);                    // mask bytes of the current group
    // (LDS is handed out in 1280-byte granules on gfx950: the 20x20, K = 2 workgroup must stay <= 32 000 bytes for five
    // workgroups per CU, the 10x10 + rotation one <= 20 480 for eight -- it is exactly 20 480)
    static constexpr int OFF_BAL = (OFF_REC + NBW * (int)sizeof(TileRec) + 7) & ~7;   // ballots of the candidate passes
    static constexpr int OFF_P = round16(OFF_BAL + (NPASS > 2 ? EPW * 2 * NPASS * 8 : 0));
    // A wave that owns ONE bin (20x20) only ever holds a ONE-word prefix image: a bin taller than kLowTop is scanned
    // in two phases, upper word then lower word, in the same 3.5 KB (8 workgroups per CU instead of 5 with a two-word image).
    static constexpr int KP = (K == 2 && EPW == 1) ? 1 : K;
    static constexpr int LDS_WAVE = OFF_P + EPW * PN * 8 * KP;  // prefix image of the current group
    static constexpr int LDS_BLOCK = kTileWaves * LDS_WAVE;
    static_assert(A % 4 == 0, "tile kernel needs W*L % 4 == 0");
    static_assert(L + 1 <= 32 && A < 2048, "slot word w1 holds nj in 5 bits and nv in 11");
    static_assert(G <= A4, "a lane group must not span more than two observation planes per pass");
    static_assert(NB <= kWave && LPB >= 1 && (EPW & (EPW - 1)) == 0 && (NIT & (NIT - 1)) == 0, "bins per workgroup");
};

// Observation and mask go out as 16-byte stores.  For the 20x20 bins they are NONTEMPORAL stores (global_store_dwordx4 ... nt):
// a lock-step of 32 768 such bins writes 263 MB, more than the Infinity Cache holds, and keeping those lines out of it is
// worth 2.5 % (one output set) to 3.8 % (outputs rotated): 56.7 -> 55.3 / 57.6 -> 55.4 us.  The 10x10 bins lose with them
// (28.4 -> 36.6 us: their 133 MB per lock-step is what the cache absorbs) -- profiles/r4zm_*, r4zn_*.
#if defined(__clang__)
typedef float bpp_v4f __attribute__((ext_vector_type(4)));
#else
typedef float bpp_v4f __attribute__((vector_size(16)));   // (the host emulator's compiler)
#endif
__device__ __forceinline__ void store_out4_nt(float4 *dst, float a, float b, float c, float d) {
    __builtin_nontemporal_store((bpp_v4f){a, b, c, d}, (bpp_v4f *)dst);
}

// The kernel itself, in its two forms (bpp_tile_body.inl).
// amdgpu_num_sgpr(80): a CU admits eight 256-thread workgroups only up to 80 scalar registers (bpp_kernels.hip: resource
// cliffs).  The kernels of the BASELINE configs sit at 77-80 by themselves; the 20x20 + rotation step kernel compiled to 84
// (seven workgroups per CU) -- with the cap the compiler parks a few launch constants in vector-register lanes instead.
#define BPP_TILE_NAME bpp_tile_kernel
#define BPP_TILE_CACHE false
#define BPP_TILE_ATTR __attribute__((amdgpu_num_sgpr(80)))
#include "bpp_tile_body.inl"
#undef BPP_TILE_NAME
#undef BPP_TILE_CACHE
#undef BPP_TILE_ATTR
// The step kernel of a launch with a row cache.  amdgpu_num_sgpr: the cache's control logic takes the body to 86-90 scalar
// registers, i.e. to seven workgroups per CU (MI355X_MICROARCH.md: eight only up to 80); held to 80 the compiler parks a
// handful of launch constants in vector-register lanes instead.
#define BPP_TILE_NAME bpp_tile_kernel_q
#define BPP_TILE_CACHE true
#define BPP_TILE_ATTR __attribute__((amdgpu_num_sgpr(80)))
#include "bpp_tile_body.inl"
#undef BPP_TILE_NAME
#undef BPP_TILE_CACHE
#undef BPP_TILE_ATTR
