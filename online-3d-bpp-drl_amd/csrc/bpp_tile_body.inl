// bpp_tile_body.inl -- the body of the tile kernel, included by bpp_tile_kernel.inl once per form:
//   BPP_TILE_NAME   kernel name          BPP_TILE_ATTR   extra function attributes
//   BPP_TILE_CACHE  false: the plain kernel (every mode).  true (step only): the launch has a bpp_batch.seq_cache (RowCache in
//                   bpp_kernels.hip) -- the look-ahead items come from the bin's cache line and the first p.ncopy workgroups
//                   are copier workgroups.
// Two kernels from one text rather than a run-time test or a shared device function: the plain step kernel sits at exactly
// 80 scalar registers (eight workgroups per CU); both forms behind a run-time test cost 6-10 more, and routing both through
// one inlined function moved the 20x20 kernel from 78 to 82.
template <int W, int L, int K, bool ROT, int MODE, int EPW, int NIT>
__global__ __launch_bounds__(kWave * kTileWaves) BPP_TILE_ATTR void BPP_TILE_NAME(const Params p) {
    constexpr bool CACHE = BPP_TILE_CACHE;
    using T = TileGeo<W, L, K, ROT, EPW, NIT>;
    constexpr int A = T::A, A4 = T::A4, M = T::M, M4 = T::M4, PW = T::PW, PN = T::PN, G = T::G, NB = T::NB, LPB = T::LPB;
    constexpr int NPASS = T::NPASS, NBW = T::NBW;
    constexpr bool BAL_REGS = NPASS <= 2;   // ballots stay in scalar registers (fully unrolled passes)
    constexpr bool kAccLate = EPW > 1;
    constexpr bool kNtOut = A >= 400 && MODE != kResetInit && MODE != kResetAdvance;   // nontemporal output stores (store_out4_nt in bpp_tile_kernel.inl; a reset's are slower with them: 40 -> 52 us)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (CACHE) {
        if ((int)blockIdx.x < p.ncopy) {               // copier workgroup: serves the refresh requests of kCopierBins bins, nothing else
            static_assert(T::LDS_WAVE >= 3 * 4 * 256, "a copier wave lists up to 256 requests in its LDS area");
            static_assert(kCopierBins == kTileWaves * 4 * kWave, "a copier workgroup's waves serve 4 x 64 bins each (row_cache_copier): every bin of the workgroup must have a wave");
            row_cache_copier(p, (int)blockIdx.x * kCopierBins + wid * (kCopierBins / kTileWaves), (uint32_t *)(smem + wid * T::LDS_WAVE));
            return;
        }
    }
    // (the mask-only entry points have no deciding wave and no workgroup barrier: their workgroups may be launched with fewer waves)
    const int wg_bins = (MODE == kMaskObs || MODE == kMaskHmap) ? (int)(blockDim.x >> 6) * NBW : NB;
    const int blk_e0 = (CACHE ? xcd_block_of((int)blockIdx.x - p.ncopy, (int)gridDim.x - p.ncopy, p.xcd_remap) : xcd_block(p.xcd_remap)) * wg_bins;   // first bin of this workgroup
    const int we0 = blk_e0 + wid * NBW;                // first bin of this wave
    const int wnenv = max(0, min(NBW, p.E - we0));     // bins of this wave (workgroup barriers below: no early return)
    const int el = lane / G, sl = lane % G;            // this lane's bin within a group, position within the bin
    unsigned char *wb = smem + wid * T::LDS_WAVE;
    uint8_t *hmw = wb;                                 // [NBW][A] byte tiles of all the wave's bins
    uint8_t *mk = wb + T::OFF_MK;
    uint32_t *mk32 = (uint32_t *)mk;
    TileRec *recw = (TileRec *)(wb + T::OFF_REC);      // [NBW]
    uint64_t *balm = (uint64_t *)(wb + T::OFF_BAL);
    Ent<K> *P = (Ent<K> *)(wb + T::OFF_P);
    const uint32_t hclamp = (uint32_t)p.H + 1u;        // heights above H all behave like H+1 (never feasible)
    constexpr int KQ = (A4 + G - 1) / G;               // tile quads per lane
    constexpr int KM = (M4 + G - 1) / G;               // mask quads per lane

    BPP_STAMP(p, 0);
    if (BPP_ABL(p, 16)) return;
    // ---- deciding wave: per-bin loads first, their latency overlaps the staging -----------------------
    const int db = lane / LPB, ql = lane % LPB;        // deciding wave: bin within the workgroup, lane within the bin
    const int dec_nb = max(0, min(NB, p.E - blk_e0));
    const bool dactive = db < dec_nb;
    const int dec_e = blk_e0 + (dactive ? db : 0);
    bpp_env_state st0;
    int64_t act0 = 0;
    uint2 ctl0 = make_uint2(0u, 0u);                    // CACHE: the bin's control word (RowCache)
    uint32_t la_ok = 0, la_f1 = 0, la_f2 = 0, outcome = 0;   // CACHE: look-ahead loads in flight; 1 placed, 2 finished, 0 neither
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    bool acc_late = false;
    if (MODE == kStep && wid == 0 && !BPP_ABL(p, 32)) {
        st0 = p.state[dec_e];
        act0 = p.actions[dec_e];
        if constexpr (CACHE) ctl0 = row_cache(p.cache, p.E).ctl[dec_e];
        if constexpr (kAccLate) {   // the row of EVERY bin, read with the state record (a prefetch: see kAccLate above)
            if (p.ep_acc != nullptr && !BPP_ABL(p, 128)) {
                const double *ea0 = (const double *)__builtin_assume_aligned(p.ep_acc + 4 * (size_t)dec_e, 32);
                acc0 = ea0[0], acc1 = ea0[1], acc2 = ea0[2], acc3 = ea0[3];
            }
        }
    }

    // ---- phase 1: stage the byte tiles of ALL the wave's bins (lane owns quads sl + G*k of bin it*EPW + el) ----
    if (MODE == kStep) {
        uint32_t v[NIT][KQ];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bool mine = it * EPW + el < wnenv;
            const uint32_t *gh = (const uint32_t *)(p.hmap + (size_t)(we0 + it * EPW + el) * A) + sl;
#pragma unroll
            for (int k = 0; k < KQ; ++k) v[it][k] = (mine && sl + G * k < A4 && !BPP_ABL(p, 64)) ? gh[G * k] : 0u;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bool mine = it * EPW + el < wnenv;
            uint32_t *hm32 = (uint32_t *)(hmw + it * EPW * A);
#pragma unroll
            for (int k = 0; k < KQ; ++k)
                if (mine && sl + G * k < A4) hm32[el * A4 + sl + G * k] = v[it][k];
        }
    } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const bool mine = it * EPW + el < wnenv;
            const int e0 = we0 + it * EPW;
            uint32_t *hm32 = (uint32_t *)(hmw + it * EPW * A);
            if (MODE == kMaskHmap) {
                const int4 *gh = (const int4 *)(p.hmap_in + (size_t)(e0 + el) * A) + sl;
#pragma unroll
                for (int k = 0; k < KQ; ++k)
                    if (mine && sl + G * k < A4) {
                        const int4 v = gh[G * k];
                        hm32[el * A4 + sl + G * k] = min((uint32_t)v.x, 255u) | (min((uint32_t)v.y, 255u) << 8) |
                                                     (min((uint32_t)v.z, 255u) << 16) | (min((uint32_t)v.w, 255u) << 24);
                    }
            } else if (MODE == kMaskObs) {
                const float4 *go = (const float4 *)(p.obs_in + (size_t)(e0 + el) * 4 * A) + sl;  // acktr/utils.py:41-47
#pragma unroll
                for (int k = 0; k < KQ; ++k)
                    if (mine && sl + G * k < A4) {
                        const float4 v = go[G * k];
                        hm32[el * A4 + sl + G * k] = min((uint32_t)(int)v.x, 255u) | (min((uint32_t)(int)v.y, 255u) << 8) |
                                                     (min((uint32_t)(int)v.z, 255u) << 16) | (min((uint32_t)(int)v.w, 255u) << 24);
                    }
            } else {
#pragma unroll
                for (int k = 0; k < KQ; ++k)
                    if (mine && sl + G * k < A4) hm32[el * A4 + sl + G * k] = 0u;  // space.py:22
            }
        }
    }
    // The mask-only entry points have no per-bin chain: every wave fetches its own bins' items and nothing crosses
    // waves -- no workgroup barrier at all (round 2 ran them through the deciding wave and both barriers).
    constexpr bool kDecide = MODE == kStep || MODE == kResetInit || MODE == kResetAdvance;
    BPP_STAMP(p, 1);
    if constexpr (kDecide) __syncthreads();  // wave 0 reads the other waves' tiles below
    else wave_sync();
    BPP_STAMP(p, 2);

    // ---- phase 2: per-bin scalar chain in wave 0, LPB lanes per bin -------------------------------------
    bool fin = false, out_ok = false, dlead = false;
    double fin_ret = 0.0, fin_ratio = 0.0;
    float out_rew = 0.0f;
    int fin_len = 0, out_boxes = 0;
    bpp_env_state st_out;
    if (kDecide && wid == 0 && !BPP_ABL(p, 32)) {
        __builtin_amdgcn_s_setprio(3);                 // the other waves of the workgroup wait for this chain
        const bool lead = dactive && ql == 0;          // the lane that writes the bin's results
        dlead = lead;
        const int e = dec_e;
        const int ow = db / NBW, oel = db % NBW;       // owning wave, bin within it
        unsigned char *ob = smem + ow * T::LDS_WAVE;
        const uint8_t *ohm = ob + oel * A;
        TileRec r;
        r.item = 0;
        r.place = 0;
        r.flags = 0;
        if (MODE == kStep) {
            bpp_env_state st = st0;
            const int64_t act = act0;
            // binCreator.py:15-18: current / next / first-of-next-episode items come from the state record;
            // the pool entries the NEXT step needs are fetched speculatively for both outcomes.
            int seq_n = st.seq + p.seq_stride;
            seq_n = seq_n >= p.P ? seq_n - p.P : seq_n;
            int seq_nn = seq_n + p.seq_stride;
            seq_nn = seq_nn >= p.P ? seq_nn - p.P : seq_nn;
            const uint32_t it_cur = st.item_cur, it_nxt = st.item_next, it_rst = st.item_reset;
            uint32_t sp_ok, sp_f1, sp_f2;
            if constexpr (CACHE) {
                // The control word says whether the current line answers this step's look-ahead; the three loads (from the
                // line, else from the ring) are only ISSUED here: their values go into the state record behind the second
                // barrier (la_ok / la_f1 / la_f2), so that not even a cache hit's latency lies on the path the other waves wait for.
                const RowCache rc = row_cache(p.cache, p.E);
                uint32_t c_cur = ctl0.x & 1u;                   // current line; pending: 1 = asked for in the previous launch, 2 = written meanwhile
                if (((ctl0.x >> 1) & 3u) == 2u) c_cur ^= 1u;    // the line asked for two launches ago was written by the last one
                const uint32_t c_tag = c_cur ? ctl0.y >> 16 : ctl0.y & 0xffffu;               // (its episode + 1) & 0xffff, 0 = no line
                const uint32_t c_c0 = (ctl0.x >> (c_cur ? 16 : 3)) & 0x1fffu;                // its first item
                const uint32_t d = ((uint32_t)st.episode + 1u - c_tag) & 0xffffu;            // rows the bin has moved on since the line was built
                const uint32_t x = (uint32_t)min(st.cursor + 2, p.T - 3);                    // item the step looks ahead to
                const bool hit = c_tag != 0u && ((d == 0u && x - c_c0 < (uint32_t)kLineItems) || (d == 1u && x < (uint32_t)(kLineNext - 2)));
                if (dactive && ql == 0) BPP_CACHE_STAT(hit);
                const LookAheadAt la = look_ahead_at(p, st.seq, seq_n, seq_nn, st.cursor);
                const uint32_t *ln = rc.lines + ((size_t)e * 2 + c_cur) * kLineWords;
                const uint32_t *a_ok = hit ? ln + (d == 0u ? 12u + x - c_c0 : 2u + x) : p.pool + la.ok;
                const uint32_t *a_f1 = hit ? ln + (d == 0u ? 8 : 0) : p.pool + la.f1;
                const uint32_t *a_f2 = hit ? ln + (d == 0u ? 9 : 1) : p.pool + la.f2;
                la_ok = *a_ok, la_f1 = *a_f1, la_f2 = *a_f2;
                sp_ok = it_nxt, sp_f1 = it_nxt, sp_f2 = it_rst;                               // (placeholders: patched behind the barrier)
            } else {
                const LookAheadAt la = look_ahead_at(p, st.seq, seq_n, seq_nn, st.cursor);
                sp_ok = p.pool[la.ok], sp_f1 = p.pool[la.f1], sp_f2 = p.pool[la.f2];
            }
            const int ix = it_cur & 255, iy = (it_cur >> 8) & 255, iz = (it_cur >> 16) & 255;
            const bool noop = act == BPP_ACTION_NOOP;                  // include/bpp_abi.h: the bin is left alone
            int64_t idx = act;                                         // bin3D.py:96-105
            const bool flag = ROT && idx > A;
            if (flag) idx -= A;
            const int x = flag ? iy : ix, y = flag ? ix : iy, z = iz;  // space.py:166-172
            bool ok = dactive && idx >= 0 && idx < (int64_t)(W + 1) * L;
            int lx = 0, ly = 0;
            if (ok) {
                lx = (int)idx / L;                                     // space.py:153-156
                ly = (int)idx - lx * L;
                ok = (lx + x <= W) && (ly + y <= L);                   // space.py:112-115
            }
            int top = 0;
            if (ok) {   // uniform over the bin's LPB lanes
                const uint8_t *hb = ohm + lx * L + ly;
                int mh = 0, ma = 0;                                    // space.py:127-129
                if (x <= 5 && y <= 5) {
                    // common item sizes: rows ql, ql + LPB, ... of the window in this lane, predicated reads
                    constexpr int NR = (5 + LPB - 1) / LPB;
                    int v[NR][5];
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                        for (int b = 0; b < 5; ++b) {
                            const int a = ql + rr * LPB;
                            v[rr][b] = (a < x && b < y) ? (int)hb[a * L + b] : -1;
                        }
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                        for (int b = 0; b < 5; ++b) mh = max(mh, v[rr][b]);
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                        for (int b = 0; b < 5; ++b) ma += (v[rr][b] == mh);
                } else {
                    for (int a = ql; a < x; a += LPB)
                        for (int b = 0; b < y; ++b) {
                            const int v = hb[a * L + b];
                            ma = v > mh ? 1 : ma + (v == mh);
                            mh = max(mh, v);
                        }
                }
                // merge (max, count) over the bin's LPB lanes; every lane ends up with the window's pair
                // (round 6: on the DPP data path -- quad_perm within a bin's four lanes, row_ror within its sixteen -- instead of
                // ds_bpermute shuffles: this merge sits in the serial chain the other three waves wait for)
                auto merge = [&](int m2, int c2) {
                    const int nm = max(mh, m2);
                    ma = (mh == nm ? ma : 0) + (m2 == nm ? c2 : 0);
                    mh = nm;
                };
                if constexpr (LPB == 4) {
                    merge(__builtin_amdgcn_update_dpp(0, mh, 0xB1, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, ma, 0xB1, 0xf, 0xf, true));   // quad_perm:[1,0,3,2]
                    merge(__builtin_amdgcn_update_dpp(0, mh, 0x4E, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, ma, 0x4E, 0xf, 0xf, true));   // quad_perm:[2,3,0,1]
                } else if constexpr (LPB == 16) {
                    merge(__builtin_amdgcn_update_dpp(0, mh, 0x128, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, ma, 0x128, 0xf, 0xf, true));  // row_ror:8
                    merge(__builtin_amdgcn_update_dpp(0, mh, 0x124, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, ma, 0x124, 0xf, 0xf, true));  // row_ror:4
                    merge(__builtin_amdgcn_update_dpp(0, mh, 0x122, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, ma, 0x122, 0xf, 0xf, true));  // row_ror:2
                    merge(__builtin_amdgcn_update_dpp(0, mh, 0x121, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, ma, 0x121, 0xf, 0xf, true));  // row_ror:1
                } else {
#pragma unroll
                    for (int d = 1; d < LPB; d <<= 1) merge(__shfl_xor(mh, d, kWave), __shfl_xor(ma, d, kWave));
                }
                const int r00 = hb[0], r10 = hb[(x - 1) * L], r01 = hb[y - 1], r11 = hb[(x - 1) * L + y - 1];
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
            out_rew = (float)rew;                                      // acktr/envs.py:192
            out_ok = ok || noop;
            out_boxes = st.n_boxes;                                    // bin3D.py:111,124
            fin = lead && !ok && !noop;
            fin_ret = st.ep_ret;
            fin_ratio = ratio;
            fin_len = st.ep_len;
            if (ok) {
                st.cursor += 1;                                        // bin3D.py:116-117
                st.item_cur = it_nxt;
                st.item_next = sp_ok;
                st.hmax = max(st.hmax, (uint32_t)top);                 // highest cell of the bin (space.py:42-45 raised the window to `top`)
                r.item = it_nxt;
                r.place = (uint32_t)lx | ((uint32_t)ly << 8) | ((uint32_t)x << 16) | ((uint32_t)y << 24);
                r.flags = 1u | ((uint32_t)top << 8) | (st.hmax <= (uint32_t)kLowTop ? 4u : 0u);
            } else if (noop) {
                r.item = it_cur;
                r.flags = st.hmax <= (uint32_t)kLowTop ? 4u : 0u;
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
                r.flags = 2u | 4u;
            }
            // written behind the second barrier.  (Loading the look-ahead pool entries there as well -- they only go into this
            // record -- instead of speculatively before the decision was measured twice, state store right after the loads
            // and state tail stored at the end of the kernel: 28.3 -> 28.9 / 29.2 us, stream mode 52.4 -> 55.3 us per lock-step.)
            st_out = st;
            if constexpr (CACHE) outcome = ok ? 1u : (noop ? 0u : 2u);
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
            if (lead) p.state[e] = st;
            if (lead && p.cache != nullptr) row_cache_drop(p, e);
            r.item = st.item_cur;
            r.flags = 2u | 4u;
        }
        if (lead) ((TileRec *)(ob + T::OFF_REC))[oel] = r;
        __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (!kDecide) {   // mask-only entry points: the owning wave reads its bins' items itself (NIT == 1)
        if (el < wnenv && sl == 0) {
            const int e = we0 + el;
            TileRec r;
            if (MODE == kMaskObs) {
                const float *o = p.obs_in + (size_t)e * 4 * A;         // acktr/utils.py:43-45
                r.item = pack_item((int)o[A], (int)o[2 * A], (int)o[3 * A]);
            } else {
                const int32_t *it = p.items_in + (size_t)e * 3;
                r.item = pack_item(it[0], it[1], it[2]);
            }
            r.place = 0;
            r.flags = 0;
            recw[el] = r;
        }
    }
    // work that does not depend on the decisions, done by the waiting waves while wave 0 decides: clear the first
    // group's mask bytes and the zero row / column of the prefix image (never written again)
    {
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (sl + G * k < M4) mk32[el * M4 + sl + G * k] = 0u;
        if constexpr (EPW > 1) {
            Ent<K> zero;
#pragma unroll
            for (int k = 0; k < K; ++k) zero.w[k] = 0;
            for (int t = lane; t < EPW * (PW + W); t += kWave) {
                const int b = t / (PW + W), r = t - b * (PW + W);
                P[b * PN + (r < PW ? r : (r - PW + 1) * PW)] = zero;
            }
        }
    }
    BPP_STAMP(p, 3);
    if constexpr (kDecide) __syncthreads();
    else wave_sync();
    BPP_STAMP(p, 4);
    if (MODE == kStep && wid == 0 && dlead) {   // per-bin outputs and the state record, off the other waves' path
        const int e = dec_e;
        // episode statistics (main.py:159-162): a finished episode is added to the bin's own accumulator row -- plain
        // read-modify-write by this lane, no atomics (see episode_acc_add); the row's loads are issued first, their
        // latency overlaps the stores below
        const bool acc = p.ep_acc != nullptr && fin && !BPP_ABL(p, 128);
        double *ea = (double *)__builtin_assume_aligned(p.ep_acc + 4 * (size_t)e, 32);
        if (acc) acc0 = ea[0], acc1 = ea[1], acc2 = ea[2], acc3 = ea[3];
        if constexpr (kAccLate) acc_late = acc;                             // consumed at the end of the kernel
        p.reward[e] = out_rew;
        p.done[e] = out_ok ? 0 : 1;
        if (p.host_reward) {        // mirrors in mapped host memory: step_wait() then only waits for the stream
            p.host_reward[e] = out_rew;
            p.host_done[e] = out_ok ? 0 : 1;
        }
        p.counter[e] = out_boxes;
        p.ratio[e] = fin_ratio;
        p.ep_ret[e] = fin_ret;
        p.ep_len[e] = fin_len;
        if constexpr (CACHE) {
            // the look-ahead items issued in the deciding phase have arrived by now
            if (outcome == 1u) st_out.item_next = la_ok;
            if (outcome == 2u) st_out.item_next = la_f1, st_out.item_reset = la_f2;
        }
        p.state[e] = st_out;
        if constexpr (CACHE) {
            // Row cache bookkeeping, off everybody's critical path.  Ask for a new line when the bin has left the newest
            // line's row or its cursor nears the end of that line's items -- unless a request is still on its way (then the
            // question comes up again next step).
            const RowCache rc = row_cache(p.cache, p.E);
            uint32_t c_cur = ctl0.x & 1u, c_pend = (ctl0.x >> 1) & 3u;
            if (c_pend == 2u) c_cur ^= 1u, c_pend = 0u;         // (the same advance as in the deciding phase)
            else if (c_pend == 1u) c_pend = 2u;
            uint32_t t0 = ctl0.y & 0xffffu, t1 = ctl0.y >> 16, c00 = (ctl0.x >> 3) & 0x1fffu, c01 = (ctl0.x >> 16) & 0x1fffu;
            const uint32_t newest = c_pend ? c_cur ^ 1u : c_cur;
            const uint32_t n_tag = newest ? t1 : t0, n_c0 = newest ? c01 : c00;
            const uint32_t tag = ((uint32_t)st_out.episode + 1u) & 0xffffu;       // 0 (one episode in 65 536): no line, the ring is read
            const bool want = tag != n_tag || (uint32_t)st_out.cursor + 4u >= n_c0 + (uint32_t)kLineItems;
            if (want && c_pend == 0u && tag != 0u) {
                const uint32_t other = c_cur ^ 1u, nc0 = (uint32_t)st_out.cursor;
                (other ? t1 : t0) = tag;
                (other ? c01 : c00) = nc0;
                c_pend = 1u;
                rc.req[e] = (unsigned long long)((uint32_t)st_out.seq | 0x80000000u) | ((unsigned long long)(nc0 | (other << 16)) << 32);
            }
            rc.ctl[e] = make_uint2(c_cur | (c_pend << 1) | (c00 << 3) | (c01 << 16), t0 | (t1 << 16));
        }
        if (!kAccLate && acc) {
            ea[0] = acc0 + fin_ret;
            ea[1] = acc1 + fin_ratio;
            ea[2] = acc2 + (double)fin_len;
            ea[3] = acc3 + 1.0;
        }
    }

    // ---- the wave's NIT groups of EPW bins, one after the other ------------------------------------------
    for (int it = 0; it < NIT; ++it) {
        const int nenv = max(0, min(EPW, wnenv - it * EPW));   // bins of this group
        if (nenv == 0) break;                                  // wave-uniform
        const int e0 = we0 + it * EPW;                         // first bin of the group
        const bool mine = el < nenv;
        uint8_t *hm = hmw + it * EPW * A;
        uint32_t *hm32 = (uint32_t *)hm;
        TileRec *rec = recw + it * EPW;
        if (it > 0) wave_sync();   // the previous group's mask bytes / prefix image have been consumed
        if (it == 0) BPP_STAMP(p, 5);

        TileRec myrec;   // this lane's bin
        myrec.item = 0;
        myrec.place = 0;
        myrec.flags = 0;
        if (mine) myrec = rec[el];
        const bool draw = MODE == kStep && p.next_action != nullptr;
        // per (bin, orientation) constants, one lane per slot (lane sl == rot of bin el), all bins of the group at once;
        // they stay in this lane's registers and are read with v_readlane by the candidate loop
        uint32_t slotw[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        constexpr bool kResets = kDecide;                      // a bin that was just reset
        constexpr bool kResetsOnly = MODE == kResetInit || MODE == kResetAdvance;   // every bin shows an empty map
        if (EPW > 1 && mine && sl < (ROT ? 2 : 1)) {           // shows an empty map: its mask is the in-range rectangle
            make_slot_words<W, L>(myrec.item, sl, kResets && (myrec.flags & 2u) != 0u, ROT, p.H, slotw);
            slotw[7] = draw ? mix32(mix32_base(p.sample_seed, p.sample_step), (uint32_t)(p.env_id_base + e0 + el)) : 0u;
        }

        if (MODE == kStep) {
            // ---- phase 2b: apply the placement (space.py:36-46: window := max_h + z), rows over the bin's lanes;
            // a finished bin restarts from an empty map -----------------------------------------------------
            if (myrec.flags & 1u) {
                const int lx = myrec.place & 255u, ly = (myrec.place >> 8) & 255u, x = (myrec.place >> 16) & 255u, y = myrec.place >> 24;
                uint8_t *hb = hm + el * A + lx * L + ly;
                const uint8_t top = (uint8_t)(myrec.flags >> 8);
                if (x <= G && y <= 5) {
                    // the footprints CUT-2 / RS sequences consist of: one row per lane, five unconditional byte stores whose
                    // offsets are clamped to the row (the surplus ones rewrite its last byte) -- no loop, no predicate per byte
                    if (sl < x) {
                        uint8_t *r = hb + sl * L;
                        const int y1 = y - 1;
                        r[0] = top, r[min(1, y1)] = top, r[min(2, y1)] = top, r[min(3, y1)] = top, r[min(4, y1)] = top;
                    }
                } else
                for (int a = sl; a < x; a += G)
                    for (int b = 0; b < y; ++b) hb[a * L + b] = top;
            }
            if (myrec.flags & 2u) {
    #pragma unroll
                for (int k = 0; k < KQ; ++k)
                    if (sl + G * k < A4) hm32[el * A4 + sl + G * k] = 0u;
            }
            wave_sync();
        }

        if (it == 0) BPP_STAMP(p, 6);
        if (MODE == kStep || MODE == kResetInit || MODE == kResetAdvance) {
            // ---- phase 3: byte heightmap (state) + float32 observation out (bin3D.py:49-66).  The bin's 4A floats
            // are A quads; lane sl owns quads sl + G*k: the plane of a quad is a compile-time property of k, except
            // in the (at most three) passes that straddle a plane boundary, where it is a compile-time lane split.
            if (mine && !BPP_ABL(p, 8)) {
                uint32_t *gh = (uint32_t *)(p.hmap + (size_t)(e0 + el) * A) + sl;
                float4 *go = (float4 *)(p.obs + (size_t)(e0 + el) * 4 * A) + sl;
                const uint32_t item = myrec.item;
                const float fx = (float)(item & 255u), fy = (float)((item >> 8) & 255u), fz = (float)((item >> 16) & 255u);
                constexpr int KO = (A + G - 1) / G;
    #pragma unroll
                for (int k = 0; k < KO; ++k) {
                    const int q = sl + G * k;                // quad within the bin's observation row
                    const int lo = (G * k) / A4, hi = min(3, (G * k + G - 1) / A4);   // folds: G, k, A4 are constants
                    if (q < A) {
                        const int pl = (lo == hi) ? lo : (q < hi * A4 ? lo : hi);
                        if (pl == 0) {
                            const uint32_t v = hm32[el * A4 + q];
                            gh[G * k] = v;
                            if constexpr (kNtOut)
                                store_out4_nt(go + G * k, (float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24));
                            else
                                go[G * k] = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u),
                                                        (float)(v >> 24));
                        } else {
                            const float f = pl == 1 ? fx : (pl == 2 ? fy : fz);
                            if constexpr (kNtOut) store_out4_nt(go + G * k, f, f, f, f);
                            else go[G * k] = make_float4(f, f, f, f);
                        }
                    }
                }
            }
            if (p.mask == nullptr) continue;
        }

        if (it == 0) BPP_STAMP(p, 7);
        // Histogram words of this group's prefix image: K, or ONE word for a 20x20 bin whose heights all fit it (kLowTop).
        // Wave-uniform: such a wave owns a single bin.
        bool low = false;
        if constexpr (K == 2 && EPW == 1) {
            if constexpr (kDecide) {
                low = ((uint32_t)__builtin_amdgcn_readfirstlane(rec[0].flags) & 4u) != 0u;   // from the state record's hmax
            } else {                                           // mask-only: the caller's heights, whatever they are
                uint32_t mx = 0;
#pragma unroll
                for (int k = 0; k < KQ; ++k)
                    if (sl + G * k < A4) {
                        const uint32_t v = hm32[sl + G * k];
                        mx = max(max(mx, v & 255u), max(max((v >> 8) & 255u, (v >> 16) & 255u), v >> 24));
                    }
                low = __ballot(mx > (uint32_t)kLowTop) == 0ull;
            }
        }
        constexpr bool kTwoPhase = K == 2 && EPW == 1;         // tall bins: two one-word scans instead of one two-word scan
        uint32_t anymask = 0;                                  // bit b: bin b of the group has a feasible position
        // PH = 0: one scan (all levels in the image); 1: first scan of a tall bin -- the image holds the UPPER word, a
        // candidate whose window reaches into it is decided now, the others get the mask byte 2; 2: second scan -- the image
        // holds the LOWER word, the candidates marked 2 are decided, ballots / draw / fallback as in a single scan.
        auto phase4 = [&](auto kk_c, auto ph_c) {
        constexpr int KK = decltype(kk_c)::value, PH = decltype(ph_c)::value;
        Ent<KK> *P = (Ent<KK> *)(wb + T::OFF_P);               // (shadows the kernel's K-word view of the same bytes)
        // ---- phase 4a: prefix image of the height-level codes ------------------------------------------------
        if (!BPP_ABL(p, 1)) {
            if constexpr (EPW == 1) {
                if (nenv > 0) build_prefix_one_bin<W, L, KK, PH>(hm, P, hclamp, lane);
            } else {
                Ent<KK> zero;   // (row 0 and column 0 of every image were cleared before the second barrier)
#pragma unroll
                for (int k = 0; k < KK; ++k) zero.w[k] = 0;
                for (int t = lane; t < nenv * W; t += kWave) {                 // running sums along each row
                    const int b = t / W, i = t - b * W;
                    const uint8_t *row = hm + b * A + i * L;
                    Ent<KK> *pr = P + b * PN + (i + 1) * PW + 1;
                    Ent<KK> s = zero;
                    uint32_t hv[L];
    #pragma unroll
                    for (int j = 0; j < L; ++j) hv[j] = row[j];
    #pragma unroll
                    for (int j = 0; j < L; ++j) {
                        const Ent<KK> c = code_of<KK>(min(hv[j], hclamp));
    #pragma unroll
                        for (int k = 0; k < KK; ++k) s.w[k] += c.w[k];
                        pr[j] = s;
                    }
                }
                wave_sync();
                for (int t = lane; t < nenv * L; t += kWave) {                 // then down each column
                    const int b = t / L, j = t - b * L;
                    Ent<KK> *pc = P + b * PN + PW + (j + 1);
                    Ent<KK> s = zero;
                    constexpr int CH = W % 10 == 0 ? 10 : (W % 5 == 0 ? 5 : 1);
                    for (int i0 = 0; i0 < W; i0 += CH) {
                        Ent<KK> v[CH];
    #pragma unroll
                        for (int i = 0; i < CH; ++i) v[i] = pc[(i0 + i) * PW];
    #pragma unroll
                        for (int i = 0; i < CH; ++i) {
    #pragma unroll
                            for (int k = 0; k < KK; ++k) s.w[k] += v[i].w[k];
                            pc[(i0 + i) * PW] = s;
                        }
                    }
                }
                wave_sync();
            }
        }

        if (it == 0) BPP_STAMP(p, 8);
        // ---- phase 4b: feasibility of every candidate position (acktr/utils.py:37-94), bin after bin ---------
        if (it > 0 && PH != 2) {   // (the first group's mask bytes were cleared before the second barrier)
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (mine && sl + G * k < M4) mk32[el * M4 + sl + G * k] = 0u;
        }
        wave_sync();
        for (int b = 0; b < (BPP_ABL(p, 2) ? 0 : nenv); ++b) {
            const Ent<KK> *Pe = P + b * PN;
            const uint8_t *he = hm + b * A;
            uint8_t *me = mk + b * M;
            uint64_t balr[2][BAL_REGS ? NPASS : 1];   // ballots of the passes (scalar registers after unrolling)
            uint32_t dec_od[2], dec_nj[2];            // per-orientation index decode, kept for the draw
            uint32_t hsh = 0;
            int tot = 0;                              // feasible candidates so far (both orientations)
#pragma unroll
            for (int rot = 0; rot < (ROT ? 2 : 1); ++rot) {                // utils.py:81-89: second half
                // the slot's constants, wave-uniform: read from the registers of the slot's lane -- or, when the wave
                // owns a single bin, straight from the item on the scalar unit
                uint32_t w0, w1, w2, w3, w4, w5, w6;
                if constexpr (EPW > 1) {
                    const int src = b * G + rot;           // lane (el = b, sl = rot)
                    w0 = (uint32_t)__builtin_amdgcn_readlane(slotw[0], src), w1 = (uint32_t)__builtin_amdgcn_readlane(slotw[1], src);
                    w2 = (uint32_t)__builtin_amdgcn_readlane(slotw[2], src), w3 = (uint32_t)__builtin_amdgcn_readlane(slotw[3], src);
                    w4 = (uint32_t)__builtin_amdgcn_readlane(slotw[4], src), w5 = (uint32_t)__builtin_amdgcn_readlane(slotw[5], src);
                    w6 = (uint32_t)__builtin_amdgcn_readlane(slotw[6], src);
                } else {
                    const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane(rec[b].item);
                    const uint32_t flg = (uint32_t)__builtin_amdgcn_readfirstlane(rec[b].flags);
                    uint32_t ww[7];
                    make_slot_words<W, L>(item, rot, kResets && (flg & 2u) != 0u, ROT, p.H, ww);
                    w0 = ww[0], w1 = ww[1], w2 = ww[2], w3 = ww[3], w4 = ww[4], w5 = ww[5], w6 = ww[6];
                    if (rot == 0 && draw) hsh = mix32(mix32_base(p.sample_seed, p.sample_step), (uint32_t)(p.env_id_base + e0 + b));
                }
                const uint32_t od = w0;
                const int nj = (int)((w1 >> 11) & 31u), nv = (int)(w1 & 0x7ffu), hz1 = (int)((w6 >> 16) & 0x1ffu);
                const bool valid = (w1 >> 28) & 1u, big = (w1 >> 29) & 1u, fresh = (w6 >> 30) & 1u, square = ((w1 & w6) >> 31) & 1u;
                const int xPW = (int)(w2 & 0xffffu), y = (int)(w2 >> 16), x = (int)((w5 >> 16) & 255u);
                const int o10 = (int)(w3 & 0xffffu), o01 = (int)(w3 >> 16);
                const int t95 = (int)(w4 & 0xffffu), t85 = (int)(w4 >> 16), t50 = (int)(w5 & 0xffffu);
                if constexpr (EPW == 1) {
                    dec_od[rot] = od;
                    dec_nj[rot] = (uint32_t)nj;
                }
#pragma unroll
                for (int ps = 0; ps < (BAL_REGS ? NPASS : 1); ++ps) balr[rot][ps] = 0ull;
                if (!BAL_REGS) {
                    for (int ps = lane; ps < NPASS; ps += kWave) balm[(b * 2 + rot) * NPASS + ps] = 0ull;
                    wave_sync();
                }
                if (PH == 1 && ROT && rot == 1 && square) continue;      // (copied from the first half by the second scan)
                if (ROT && rot == 1 && square) {
                    // square footprint: the turned item's mask (utils.py:81-89) equals the first half
#pragma unroll
                    for (int k = 0; k < (A4 + kWave - 1) / kWave; ++k)
                        if (lane + kWave * k < A4) mk32[b * M4 + A4 + lane + kWave * k] = mk32[b * M4 + lane + kWave * k];
                    if (BAL_REGS) {
#pragma unroll
                        for (int ps = 0; ps < (BAL_REGS ? NPASS : 1); ++ps) balr[1][ps] = balr[0][ps];
                    } else {
                        wave_sync();
                        for (int ps = lane; ps < NPASS; ps += kWave) balm[(b * 2 + 1) * NPASS + ps] = balm[(b * 2) * NPASS + ps];
                    }
                    tot += tot;
                    continue;
                }
                if (!valid) continue;                                       // item does not fit at all
                // One candidate loop per case, so that no bin-uniform condition is re-tested per candidate.  YC != 0: the
                // loop compiled for ONE item length y -- the y-offsets of the prefix-image and corner reads are then
                // immediates (two address additions instead of six) and the index decode is a literal.  Used where a wave
                // owns ONE bin and runs five or six passes through the same loop (20x20: 61.4 -> 58.7 us); with four bins per
                // wave every wave hops between the variants and the instruction cache loses more than the additions cost
                // (10x10: 28.3 -> 28.8 us, + rotation 34.8 -> 36.7 us; one loop per footprint x * y, sixteen variants with
                // every offset immediate, 19 -> 32 KB of code for the rotation kernel: 37.7 -> 40.9 us).  The lengths 2..5
                // are what CUT-2 / RS sequences consist of (acktr/arguments.py:122-128); anything else runs the YC = 0 form.
                auto run = [&](auto big_c, auto empty_c, auto yc_c) {
                    constexpr bool BIG = decltype(big_c)::value, EMPTY = decltype(empty_c)::value;
                    constexpr int YC = decltype(yc_c)::value;
                    constexpr bool SP = YC != 0;
                    constexpr int cNJ = SP ? L - YC + 1 : 1;
                    constexpr uint32_t cOD = ((1u << kCandShift) + (uint32_t)cNJ - 1u) / (uint32_t)cNJ;
                    const int c_nv = nv, c_nj = SP ? cNJ : nj, c_y = SP ? YC : y, c_xPW = xPW;
                    const uint32_t c_od = SP ? cOD : od;
                    const int c_o10 = o10, c_o01 = SP ? YC - 1 : o01;
                    const int c_t95 = t95, c_t85 = t85, c_t50 = t50;
    #pragma unroll(BAL_REGS ? NPASS : 1)
                    for (int ps = 0; ps < NPASS; ++ps) {
                        if (ps * kWave >= c_nv) break;                      // wave-uniform (compile-time when SP)
                        const int t = lane + ps * kWave;
                        bool f = false;
                        if (t < c_nv) {
                            const int i = (int)(((uint32_t)t * c_od) >> kCandShift), j = t - i * c_nj;
                            uint8_t *mb = me + rot * A + i * L + j;
                            bool need = true;                      // second scan: only what the first one left open
                            if (PH == 2) {
                                const uint32_t prev = *mb;
                                need = prev == 2u;
                                f = prev == 1u;
                            }
                            if (EMPTY) {
                                f = hz1 > 0;  // empty map: max_h = 0 over the whole window, every in-range position passes
                                *mb = f ? 1 : 0;
                            } else if (need) {
                                const Ent<KK> *Pb = Pe + i * PW + j;
                                int mh, ma;
                                bool decided = true;
                                if (!BIG) {
                                    const Ent<KK> a = Pb[0], bb = Pb[c_y], cc = Pb[c_xPW], d = Pb[c_xPW + c_y];
                                    Ent<KK> h;
    #pragma unroll
                                    for (int k = 0; k < KK; ++k) h.w[k] = (a.w[k] + d.w[k]) - (bb.w[k] + cc.w[k]);
                                    top_of<KK>(h, mh, ma);
                                    if (PH == 1) decided = h.w[0] != 0ull;   // some cell of the window lies in the upper word
                                    // (leaving a pass of the first scan early when none of its windows reaches the upper
                                    // word was measured: 56.8 vs 56.7 us -- nothing)
                                } else {
                                    window_top<KK, PH != 0>(Pe, PW, i, j, x, y, mh, ma);
                                    if (PH == 1) decided = mh >= 0;
                                }
                                if (PH == 1) mh += kLevelsPerWord;
                                const uint8_t *hb = he + i * L + j;
                                const int r00 = hb[0], r10 = hb[c_o10], r01 = hb[c_o01], r11 = hb[c_o10 + c_o01];
                                // utils.py:23-33 on lane masks: all four corners at max_h -> t50, exactly three -> t85
                                const bool e0c = r00 == mh, e1c = r10 == mh, e2c = r01 == mh, e3c = r11 == mh;
                                const bool a01 = e0c && e1c, o01c = e0c || e1c, a23 = e2c && e3c, o23 = e2c || e3c;
                                const bool all4 = a01 && a23, ge3 = (a01 && o23) || (a23 && o01c);
                                const int thr = all4 ? c_t50 : (ge3 ? c_t85 : c_t95);
                                f = (mh < hz1) && (ma >= thr);                 // utils.py:20-33
                                if (p.rule == BPP_RULE_SPACE) {                // space.py:122-125: sc >= 3
                                    const int rm = max(max(r00, r10), max(r01, r11));
                                    f = f && ((r00 == rm) + (r10 == rm) + (r01 == rm) + (r11 == rm) >= 3);
                                }
                                *mb = (PH == 1 && !decided) ? 2 : (f ? 1 : 0);
                            }
                        }
                        if (PH == 1) continue;                     // no ballots before every candidate is decided
                        const unsigned long long bl = __ballot(f);
                        tot += __popcll(bl);
                        if (BAL_REGS) balr[rot][ps] = bl;
                        else if (lane == 0) balm[(b * 2 + rot) * NPASS + ps] = bl;
                    }
                };
                using TT = std::true_type;
                using FF = std::false_type;
                using I0 = std::integral_constant<int, 0>;
                // length-specialised loops: the 20x20x20 step and mask kernels (resets only see empty maps)
                constexpr bool kSpec = EPW == 1 && W == 20 && L == 20 && K == 2 && !kResetsOnly;
                if (fresh) {
                    run(FF{}, TT{}, I0{});
                } else if (big) {
                    run(TT{}, FF{}, I0{});
                } else if (kSpec && y == 2) {
                    run(FF{}, FF{}, std::integral_constant<int, kSpec ? 2 : 0>{});
                } else if (kSpec && y == 3) {
                    run(FF{}, FF{}, std::integral_constant<int, kSpec ? 3 : 0>{});
                } else if (kSpec && y == 4) {
                    run(FF{}, FF{}, std::integral_constant<int, kSpec ? 4 : 0>{});
                } else if (kSpec && y == 5) {
                    run(FF{}, FF{}, std::integral_constant<int, kSpec ? 5 : 0>{});
                } else {
                    run(FF{}, FF{}, I0{});
                }
                wave_sync();   // reconvergence point of the candidate loop (also orders the LDS ballots)
            }
            if (PH == 1) continue;                                 // fallback flag and draw belong to the second scan
            anymask |= tot > 0 ? 1u << b : 0u;

            // ---- phase 4c (optional): draw the next action uniformly among the feasible entries -------------
            // Same result as bpp_sample_feasible on the mask this step writes: pick = hash * count >> 32, the
            // pick-th set entry in index order = the pick-th set ballot bit in pass order (candidates are
            // enumerated in index order, first orientation first); all-ones fallback: pick among all M entries.
            if (draw) {
                const int e = e0 + b;
                if constexpr (EPW > 1) hsh = (uint32_t)__builtin_amdgcn_readlane(slotw[7], b * G);
                if (tot == 0) {
                    if (lane == 0) p.next_action[e] = (int64_t)__umulhi(hsh, (uint32_t)M);
                } else {
                    // the pick-th set ballot bit, passes in enumeration order.  (A branch-free form of this search --
                    // scalar selects of the winning (orientation, pass, ballot) -- was measured: 37.5 instead of 34.6 us for
                    // the rotation kernel; the 64-bit selects and their wait states cost more than the taken branch.)
                    int rem = (int)__umulhi(hsh, (uint32_t)tot);
                    bool found = false;
    #pragma unroll
                    for (int rot = 0; rot < (ROT ? 2 : 1); ++rot) {
    #pragma unroll(BAL_REGS ? NPASS : 1)
                        for (int ps = 0; ps < NPASS; ++ps) {
                            unsigned long long bl;
                            if (BAL_REGS) {
                                bl = balr[rot][ps];
                            } else {
                                const uint64_t v = balm[(b * 2 + rot) * NPASS + ps];
                                // (the builtin returns a signed int: go through uint32_t, or the low half sign-extends)
                                const uint32_t vhi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
                                const uint32_t vlo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
                                bl = ((unsigned long long)vhi << 32) | (unsigned long long)vlo;
                            }
                            const int c = __popcll(bl);
                            if (!found && rem < c) {
                                found = true;
                                // index decode of this orientation: the slot's first word again
                                uint32_t od, nj;
                                if constexpr (EPW > 1) {
                                    od = (uint32_t)__builtin_amdgcn_readlane(slotw[0], b * G + rot);
                                    nj = ((uint32_t)__builtin_amdgcn_readlane(slotw[1], b * G + rot) >> 11) & 31u;
                                } else {
                                    od = dec_od[rot], nj = dec_nj[rot];
                                }
                                const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(bl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bl, 0u));
                                if (((bl >> lane) & 1ull) && (int)below == rem) {
                                    const int t = lane + ps * kWave;
                                    const int i = (int)(((uint32_t)t * od) >> kCandShift), j = t - i * (int)nj;
                                    p.next_action[e] = (int64_t)(rot * A + i * L + j);
                                }
                            }
                            rem -= found ? 0 : c;
                        }
                    }
                }
            }
        }
        wave_sync();
        };   // phase4
        using I0_ = std::integral_constant<int, 0>;
        if constexpr (kTwoPhase) {
            if (low) {
                phase4(std::integral_constant<int, 1>{}, I0_{});
            } else {
                phase4(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
                phase4(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
            }
        } else {
            phase4(std::integral_constant<int, K>{}, I0_{});
        }

        if (it == 0) BPP_STAMP(p, 9);
        // ---- phase 5: float32 mask out, all-ones fallback (utils.py:59-60,91-92) ---------------------------
        if (mine && !BPP_ABL(p, 4)) {
            float4 *gm = (float4 *)(p.mask + (size_t)(e0 + el) * M) + sl;
            const bool anyf = ((anymask >> el) & 1u) != 0u;
    #pragma unroll
            for (int k = 0; k < KM; ++k)
                if (sl + G * k < M4) {
                    const uint32_t v = anyf ? mk32[el * M4 + sl + G * k] : 0x01010101u;
                    if constexpr (kNtOut)
                        store_out4_nt(gm + G * k, (float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24));
                    else
                        gm[G * k] = make_float4((float)(v & 255u), (float)((v >> 8) & 255u), (float)((v >> 16) & 255u), (float)(v >> 24));
                }
        }
        if (it == NIT - 1) BPP_STAMP(p, 10);
    }
    if (kAccLate && MODE == kStep && wid == 0 && acc_late) {   // the row read behind the second barrier has long arrived
        double *ea = (double *)__builtin_assume_aligned(p.ep_acc + 4 * (size_t)dec_e, 32);
        ea[0] = acc0 + fin_ret;
        ea[1] = acc1 + fin_ratio;
        ea[2] = acc2 + (double)fin_len;
        ea[3] = acc3 + 1.0;
    }
}
