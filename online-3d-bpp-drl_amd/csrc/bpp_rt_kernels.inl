// bpp_rt_kernels.inl -- the two step / reset / mask kernels with RUNTIME geometry, included by bpp_kernels.hip inside its anonymous
// namespace (round 6: moved out of that file unchanged): bpp_kernel (cell scan: any W*L <= 1024) and bpp_fast_kernel (packed-histogram
// prefix image: W*L % 4 == 0, H <= 22), plus the prefix-image primitives (Ent, code_of, top_of, window_top, build_prefix_one_bin) that
// the compile-time-geometry kernels of bpp_tile_kernel.inl share.  Params, the helpers above them and the host side: bpp_kernels.hip.

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
// lanes scan concurrently; each lane scans its <= CS cells, the segment totals are handed up lane by lane
// (wave_shr:1 on the DPP data path) and added as offsets.  Same for the column pass.  (With lane-per-row scans only W of 64 lanes
// worked and this phase was 27 % of the 20^3 step.)
// the entry of the lane below (lane 0: zero) on the DPP data path: wave_shr:1, two 32-bit halves per word
template <int K>
__device__ __forceinline__ Ent<K> lane_below_ent(const Ent<K> &v) {
    Ent<K> r;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v.w[k], 0x138, 0xf, 0xf, true);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v.w[k] >> 32), 0x138, 0xf, 0xf, true);
        r.w[k] = ((uint64_t)hi << 32) | lo;
    }
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
        Ent<K> t = run;
#pragma unroll
        for (int d = 1; d < SR; ++d) {
            t = lane_below_ent<K>(t);          // the segment total d lanes below (round 6: DPP instead of ds_bpermute)
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
        Ent<K> t = run;
#pragma unroll
        for (int d = 1; d < SC; ++d) {
            t = lane_below_ent<K>(t);          // the segment total d lanes below (round 6: DPP instead of ds_bpermute)
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

