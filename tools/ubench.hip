// tools/ubench.hip -- instruction-cost micro-benchmarks for gfx950, used to price the step kernel's phases
// (DESIGN.md section 3.1).  Not part of the product.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench tools/ubench.hip
//
// Every test launches 256 CUs x `wps` waves per SIMD (256-thread workgroups = one wave per SIMD), each wave
// issuing ITER x 64 copies of one instruction on 8 independent register chains; reported: ns per wave-instruction
// per SIMD (time x / (wps x ITER x 64)) and the same relative to v_add_u32.  Also: calibration kernels for the
// rocprofv3 FETCH_SIZE / WRITE_SIZE counters in the step kernel's own access widths (MI355X_MICROARCH.md, HBM).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int ITER = 2000;

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP64(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S) REP8(S)

// 32-bit VALU op with two sources: v[k] = op(v[k], c)
#define DEF_VOP2(NAME, ASM)                                                          \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint32_t c0) {    \
        uint32_t v[8];                                                               \
        for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 8 + k + c0;                 \
        uint32_t c = c0 | 3u;                                                        \
        for (int it = 0; it < ITER; ++it) {                                          \
            REP64(NAME##_STEP)                                                       \
        }                                                                            \
        uint32_t s = 0;                                                              \
        for (int k = 0; k < 8; ++k) s ^= v[k];                                       \
        if (s == 0x12345u) out[threadIdx.x] = s;                                     \
    }

#define add_u32_STEP(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(add_u32, "")
#define mul_lo_STEP(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(mul_lo, "")
#define mul_hi_STEP(k) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(mul_hi, "")
#define mul_u24_STEP(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(mul_u24, "")
#define mad_u24_STEP(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(mad_u24, "")
#define lshl_add_STEP(k) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(lshl_add, "")
#define add3_STEP(k) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(add3, "")
#define cndmask_STEP(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(c) : );
DEF_VOP2(cndmask, "")
// the forms the compiler emits in the candidate loop: mask in an SGPR pair (e64), inline constants / registers as sources
#define cndmask64c_STEP(k) asm volatile("v_cndmask_b32_e64 %0, 0, 1, s[20:21]" : "=v"(v[k]) : : "s20", "s21");
DEF_VOP2(cndmask64c, "")
#define cndmask64v_STEP(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[k]) : "v"(c) : "s20", "s21");
DEF_VOP2(cndmask64v, "")
#define cmp_STEP(k) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(v[k]), "v"(c) : "vcc");
DEF_VOP2(cmp, "")
#define cvt_ubyte_STEP(k) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(v[k]));
DEF_VOP2(cvt_ubyte, "")
#define ffbh_STEP(k) asm volatile("v_ffbh_u32 %0, %0" : "+v"(v[k]));
DEF_VOP2(ffbh, "")
#define bfe_STEP(k) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(v[k]));
DEF_VOP2(bfe, "")
#define perm_STEP(k) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(perm, "")
#define mov_dpp_STEP(k) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[k]));
DEF_VOP2(mov_dpp, "")
#define add_dpp_STEP(k) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[k]));
DEF_VOP2(add_dpp, "")
#define readlane_STEP(k) asm volatile("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0" : "+v"(v[k]) : : "s20");
DEF_VOP2(readlane, "")
#define fma_f32_STEP(k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(fma_f32, "")
#define pk_add_u16_STEP(k) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(pk_add_u16, "")
#define sad_u8_STEP(k) asm volatile("v_sad_u8 %0, %0, %1, %1" : "+v"(v[k]) : "v"(c));
DEF_VOP2(sad_u8, "")
#define salu_STEP(k) asm volatile("s_add_u32 s20, s20, s21\n s_lshl_b32 s22, s22, 1" : : : "s20", "s21", "s22", "scc");
DEF_VOP2(salu, "")

// 64-bit ops on register pairs
#define DEF_V64(NAME)                                                                \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint32_t c0) {    \
        uint64_t v[8];                                                               \
        for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 8 + k + c0;                 \
        uint64_t c = ((uint64_t)c0 << 32) | 3u;                                      \
        uint32_t sh = (c0 & 3u) + 1u;                                                \
        for (int it = 0; it < ITER; ++it) {                                          \
            REP64(NAME##_STEP)                                                       \
        }                                                                            \
        uint64_t s = 0;                                                              \
        for (int k = 0; k < 8; ++k) s ^= v[k];                                       \
        if (s == 0x12345u) out[threadIdx.x] = (uint32_t)s + sh;                      \
    }
#define lshl_add_u64_STEP(k) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(v[k]) : "v"(c));
DEF_V64(lshl_add_u64)
#define lshlrev_b64_STEP(k) asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(v[k]) : "v"(sh));
DEF_V64(lshlrev_b64)
#define lshrrev_b64_STEP(k) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(v[k]) : "v"(sh));
DEF_V64(lshrrev_b64)
#define add_f64_STEP(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[k]) : "v"(c));
DEF_V64(add_f64)
#define addc_pair_STEP(k) \
    asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(*(uint32_t *)&v[k]) : "v"((uint32_t)c) : "vcc");
DEF_V64(addc_pair)

// LDS ops: address pattern = lane-linear (conflict-free) unless stated
#define DEF_LDS(NAME, SETUP)                                                          \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, uint32_t c0) {     \
        __shared__ uint64_t lds[4096];                                                \
        for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i + c0;                \
        __syncthreads();                                                              \
        const uint32_t lane = threadIdx.x & 63u;                                      \
        uint32_t addr = SETUP;                                                        \
        uint64_t v[8];                                                                \
        for (int k = 0; k < 8; ++k) v[k] = k;                                         \
        for (int it = 0; it < ITER / 4; ++it) {                                       \
            REP64(NAME##_STEP)                                                        \
            asm volatile("s_waitcnt lgkmcnt(0)");                                     \
        }                                                                             \
        uint64_t s = 0;                                                               \
        for (int k = 0; k < 8; ++k) s ^= v[k];                                        \
        if (s == 0x12345u) out[threadIdx.x] = (uint32_t)s;                            \
    }
#define ds_read_b64_STEP(k) asm volatile("ds_read_b64 %0, %1 offset:" #k "*512" : "=v"(v[k]) : "v"(addr));
DEF_LDS(ds_read_b64, (threadIdx.x >> 6) * 8192 + lane * 8)
#define ds_read_b32_STEP(k) asm volatile("ds_read_b32 %0, %1 offset:" #k "*256" : "=v"(*(uint32_t *)&v[k]) : "v"(addr));
DEF_LDS(ds_read_b32, (threadIdx.x >> 6) * 8192 + lane * 4)
#define ds_read_u8_STEP(k) asm volatile("ds_read_u8 %0, %1 offset:" #k "*64" : "=v"(*(uint32_t *)&v[k]) : "v"(addr));
DEF_LDS(ds_read_u8, (threadIdx.x >> 6) * 8192 + lane)
#define ds_read_b64_s88_STEP(k) asm volatile("ds_read_b64 %0, %1 offset:" #k "*8" : "=v"(v[k]) : "v"(addr));
DEF_LDS(ds_read_b64_s88, (threadIdx.x >> 6) * 8192 + (lane / 8) * 88 + (lane % 8) * 8)  // 11-entry row pitch like the 10x10 prefix image
#define ds_write_b8_STEP(k) asm volatile("ds_write_b8 %1, %0 offset:" #k "*64" : : "v"(*(uint32_t *)&v[k]), "v"(addr));
DEF_LDS(ds_write_b8, (threadIdx.x >> 6) * 8192 + lane)
#define ds_write_b64_STEP(k) asm volatile("ds_write_b64 %1, %0 offset:" #k "*512" : : "v"(v[k]), "v"(addr));
DEF_LDS(ds_write_b64, (threadIdx.x >> 6) * 8192 + lane * 8)
#define ds_bpermute_STEP(k) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(*(uint32_t *)&v[k]) : "v"(addr));
DEF_LDS(ds_bpermute, ((lane + 1) & 63u) * 4)

// ---- does a VALU instruction with few active lanes cost less issue time? (round 5: removing ~60 vector instructions that ran
// with 4 of 64 lanes active moved the mask kernel by 1 %): v_mul_u32_u24 under a fixed EXEC mask --------------------------
template <uint32_t LO, uint32_t HI>
__global__ __launch_bounds__(256) void k_exec_mask(uint32_t *out, uint32_t c0) {
    uint32_t v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 8 + k + c0;
    uint32_t c = c0 | 3u;
    asm volatile("s_mov_b32 exec_lo, %0\n s_mov_b32 exec_hi, %1" : : "s"(LO), "s"(HI) : "exec");
    for (int it = 0; it < ITER; ++it) {
        REP64(mul_u24_STEP)
    }
    asm volatile("s_mov_b64 exec, -1" : : : "exec");
    uint32_t s = 0;
    for (int k = 0; k < 8; ++k) s ^= v[k];
    if (s == 0x12345u) out[threadIdx.x] = s;
}
// the same with v_add_u32 (the one instruction that issues in ~2.5 cycles) and with a v_cmp + v_cndmask pair
template <uint32_t LO, uint32_t HI>
__global__ __launch_bounds__(256) void k_exec_mask_add(uint32_t *out, uint32_t c0) {
    uint32_t v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 8 + k + c0;
    uint32_t c = c0 | 3u;
    asm volatile("s_mov_b32 exec_lo, %0\n s_mov_b32 exec_hi, %1" : : "s"(LO), "s"(HI) : "exec");
    for (int it = 0; it < ITER; ++it) {
        REP64(add_u32_STEP)
    }
    asm volatile("s_mov_b64 exec, -1" : : : "exec");
    uint32_t s = 0;
    for (int k = 0; k < 8; ++k) s ^= v[k];
    if (s == 0x12345u) out[threadIdx.x] = s;
}

// which instructions pay for a sparse EXEC mask (<= 8 active lanes): every VOP2 test above, once with all lanes, once with lane 0
#define DEF_SPARSE(NAME)                                                                  \
    template <uint32_t LO, uint32_t HI>                                                   \
    __global__ __launch_bounds__(256) void ks_##NAME(uint32_t *out, uint32_t c0) {        \
        uint32_t v[8];                                                                    \
        for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 8 + k + c0;                      \
        uint32_t c = c0 | 3u;                                                             \
        asm volatile("s_mov_b32 exec_lo, %0\n s_mov_b32 exec_hi, %1" : : "s"(LO), "s"(HI) : "exec"); \
        for (int it = 0; it < ITER; ++it) {                                               \
            REP64(NAME##_STEP)                                                            \
        }                                                                                 \
        asm volatile("s_mov_b64 exec, -1" : : : "exec");                                  \
        uint32_t s = 0;                                                                   \
        for (int k = 0; k < 8; ++k) s ^= v[k];                                            \
        if (s == 0x12345u) out[threadIdx.x] = s;                                          \
    }
#define lshlrev_STEP(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[k]));
#define and_STEP(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
#define or3_STEP(k) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(v[k]) : "v"(c));
#define mov_STEP(k) asm volatile("v_mov_b32 %0, %1" : "=v"(v[k]) : "v"(c));
#define max_STEP(k) asm volatile("v_max_u32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
#define sub_STEP(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
#define xor_STEP(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
#define cvt_f32_u32_STEP(k) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[k]));
#define add_f32_STEP(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(c));
DEF_SPARSE(add_u32) DEF_SPARSE(mul_u24) DEF_SPARSE(mul_lo) DEF_SPARSE(mad_u24) DEF_SPARSE(lshl_add) DEF_SPARSE(add3) DEF_SPARSE(cndmask64v)
DEF_SPARSE(cmp) DEF_SPARSE(cvt_ubyte) DEF_SPARSE(ffbh) DEF_SPARSE(bfe) DEF_SPARSE(perm) DEF_SPARSE(fma_f32) DEF_SPARSE(lshlrev) DEF_SPARSE(and)
DEF_SPARSE(or3) DEF_SPARSE(mov) DEF_SPARSE(max) DEF_SPARSE(sub) DEF_SPARSE(xor) DEF_SPARSE(cvt_f32_u32) DEF_SPARSE(add_f32) DEF_SPARSE(readlane)

// ---- counter calibration: known byte counts in the step kernel's access widths -------------------
__global__ __launch_bounds__(256) void k_read4(const uint32_t *in, uint32_t *out, size_t n) {  // 4 B per lane loads
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s ^= in[i];
    if (s == 0x12345u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_read16(const uint4 *in, uint32_t *out, size_t n) {  // 16 B per lane loads
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        s ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (s == 0x12345u) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write16(uint4 *out, size_t n, uint32_t c) {  // 16 B per lane stores
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint4(c, c + 1, c + 2, (uint32_t)i);
}
__global__ __launch_bounds__(256) void k_write4(uint32_t *out, size_t n, uint32_t c) {  // 4 B per lane stores
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = c + (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_write8(uint2 *out, size_t n, uint32_t c) {  // 8 B per lane stores
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_uint2(c, (uint32_t)i);
}
__global__ __launch_bounds__(256) void k_write16_nt(uint4 *out, size_t n, uint32_t c) {  // 16 B per lane, nontemporal
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t *q = (uint32_t *)(out + i);
        __builtin_nontemporal_store(c, q);
        __builtin_nontemporal_store(c + 1, q + 1);
        __builtin_nontemporal_store(c + 2, q + 2);
        __builtin_nontemporal_store((uint32_t)i, q + 3);
    }
}
// the same bytes, every workgroup its own contiguous chunk (the step kernel's pattern: a workgroup writes its bins' rows)
__global__ __launch_bounds__(256) void k_write16_chunk(uint4 *out, size_t n, uint32_t c) {
    const size_t per = n / gridDim.x;
    uint4 *o = out + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) o[i] = make_uint4(c, c + 1, c + 2, (uint32_t)i);
}
__global__ __launch_bounds__(256) void k_write4_chunk(uint32_t *out, size_t n, uint32_t c) {
    const size_t per = n / gridDim.x;
    uint32_t *o = out + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) o[i] = c + (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_copy16(const uint4 *in, uint4 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

// the step kernel's NARROW reads (round 5): what does FETCH_SIZE report for them?
// one 4-byte load per lane, every lane its own line `stride` bytes apart (the speculative pool look-aheads: a scattered dword per bin)
__global__ __launch_bounds__(256) void k_read4_strided(const uint32_t *in, uint32_t *out, size_t nlines, int stride_words) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = 0;
    for (; i < nlines; i += (size_t)gridDim.x * blockDim.x) s ^= in[i * (size_t)stride_words];
    if (s == 0x12345u) out[0] = s;
}
// 48-byte array-of-structs records, one record per lane as three 16-byte loads (bpp_env_state)
__global__ __launch_bounds__(256) void k_read48_aos(const uint4 *in, uint32_t *out, size_t nrec) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t s = 0;
    for (; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
        s ^= a.x ^ b.y ^ c.z;
    }
    if (s == 0x12345u) out[0] = s;
}
// one record per FOUR lanes (the deciding wave: a bin's lead lane reads its record, the other three lanes of the bin idle)
__global__ __launch_bounds__(256) void k_read48_aos_lead(const uint4 *in, uint32_t *out, size_t nrec) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 4;
    const bool lead = (threadIdx.x & 3) == 0;
    uint32_t s = 0;
    for (; i < nrec; i += (size_t)gridDim.x * blockDim.x / 4)
        if (lead) {
            const uint4 a = in[3 * i], b = in[3 * i + 1], c = in[3 * i + 2];
            s ^= a.x ^ b.y ^ c.z;
        }
    if (s == 0x12345u) out[0] = s;
}
// 100-byte tiles read as dwords by 16-lane groups (lane sl reads dwords sl and sl + 16 of its bin's tile: bpp_tile_kernel's staging)
__global__ __launch_bounds__(256) void k_read_tile100(const uint32_t *in, uint32_t *out, size_t ntiles) {
    const int sl = threadIdx.x & 15;
    size_t b = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 16;
    uint32_t s = 0;
    for (; b < ntiles; b += (size_t)gridDim.x * blockDim.x / 16) {
        const uint32_t *t = in + b * 25;
        s ^= t[sl];
        if (sl + 16 < 25) s ^= t[sl + 16];
    }
    if (s == 0x12345u) out[0] = s;
}
// 8-byte loads, one per FOUR lanes (the deciding wave's actions)
__global__ __launch_bounds__(256) void k_read8_lead(const uint2 *in, uint32_t *out, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 4;
    const bool lead = (threadIdx.x & 3) == 0;
    uint32_t s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x / 4)
        if (lead) s ^= in[i].x;
    if (s == 0x12345u) out[0] = s;
}

template <typename K>
static double time_kernel(K kern, int wps, uint32_t *dout, int reps = 3) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int r = 0; r < reps + 1; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, dout, 1u);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv) {
    const bool calib = argc > 1 && !strcmp(argv[1], "calib");
    uint32_t *dout;
    CHECK(hipMalloc(&dout, 1 << 20));
    if (calib) {
        // run under: rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE) --kernel-trace -- tools/ubench calib
        const size_t bytes = (size_t)1 << 30;  // 1 GiB: far beyond the 256 MiB Infinity Cache
        uint4 *a, *b;
        CHECK(hipMalloc(&a, bytes));
        CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 1, bytes));
        CHECK(hipMemset(b, 2, bytes));
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        auto run = [&](const char *name, auto launch) {
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CHECK(hipEventRecord(e0));
                launch();
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("{\"calib\": \"%s\", \"bytes\": %zu, \"ms\": %.4f, \"GBps\": %.1f}\n", name, bytes, best, bytes / best / 1e6);
        };
        const int grid = 256 * 32;
        run("read4", [&] { hipLaunchKernelGGL(k_read4, dim3(grid), dim3(256), 0, 0, (const uint32_t *)a, dout, bytes / 4); });
        run("read16", [&] { hipLaunchKernelGGL(k_read16, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, dout, bytes / 16); });
        run("write4", [&] { hipLaunchKernelGGL(k_write4, dim3(grid), dim3(256), 0, 0, (uint32_t *)b, bytes / 4, 7u); });
        run("write16", [&] { hipLaunchKernelGGL(k_write16, dim3(grid), dim3(256), 0, 0, b, bytes / 16, 7u); });
        run("write8", [&] { hipLaunchKernelGGL(k_write8, dim3(grid), dim3(256), 0, 0, (uint2 *)b, bytes / 8, 7u); });
        run("write16_nontemporal", [&] { hipLaunchKernelGGL(k_write16_nt, dim3(grid), dim3(256), 0, 0, b, bytes / 16, 7u); });
        run("write16_chunk_per_workgroup", [&] { hipLaunchKernelGGL(k_write16_chunk, dim3(grid), dim3(256), 0, 0, b, bytes / 16, 7u); });
        run("write4_chunk_per_workgroup", [&] { hipLaunchKernelGGL(k_write4_chunk, dim3(grid), dim3(256), 0, 0, (uint32_t *)b, bytes / 4, 7u); });
        run("write16_chunk_4096_workgroups", [&] { hipLaunchKernelGGL(k_write16_chunk, dim3(4096), dim3(256), 0, 0, b, bytes / 16, 7u); });
        run("copy16_half", [&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, b, bytes / 32); });
        // narrow reads: `bytes` in the line = USEFUL bytes (what the lanes ask for); the lines they touch are stated in the name
        auto run_n = [&](const char *name, size_t useful, auto launch) {
            float best = 1e30f;
            for (int r = 0; r < 3; ++r) {
                CHECK(hipEventRecord(e0));
                launch();
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("{\"calib\": \"%s\", \"useful_bytes\": %zu, \"ms\": %.4f, \"useful_GBps\": %.1f}\n", name, useful, best, useful / best / 1e6);
        };
        for (int stride : {64, 128, 256}) {
            const size_t nl = bytes / stride;
            char nm[96];
            snprintf(nm, sizeof nm, "read4_one_dword_per_%dB_line (%zu lines)", stride, nl);
            run_n(nm, nl * 4, [&] { hipLaunchKernelGGL(k_read4_strided, dim3(grid), dim3(256), 0, 0, (const uint32_t *)a, dout, nl, stride / 4); });
        }
        run_n("read48_aos_record_per_lane (1 GiB of records)", bytes / 48 * 48, [&] { hipLaunchKernelGGL(k_read48_aos, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, dout, bytes / 48); });
        run_n("read48_aos_record_per_4_lanes (1 GiB of records)", bytes / 48 * 48, [&] { hipLaunchKernelGGL(k_read48_aos_lead, dim3(grid), dim3(256), 0, 0, (const uint4 *)a, dout, bytes / 48); });
        run_n("read_tile100_dwords_by_16_lane_groups (1 GiB of tiles)", bytes / 100 * 100, [&] { hipLaunchKernelGGL(k_read_tile100, dim3(grid), dim3(256), 0, 0, (const uint32_t *)a, dout, bytes / 100); });
        run_n("read8_one_per_4_lanes (1 GiB)", bytes, [&] { hipLaunchKernelGGL(k_read8_lead, dim3(grid), dim3(256), 0, 0, (const uint2 *)a, dout, bytes / 8); });
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "exec")) {
#define XM(LO, HI) {#LO " " #HI, k_exec_mask<LO, HI>, k_exec_mask_add<LO, HI>}
        struct X { const char *name; void (*k)(uint32_t *, uint32_t); void (*ka)(uint32_t *, uint32_t); };
        const X xs[] = {XM(0xffffffffu, 0xffffffffu), XM(0xffffffffu, 0u), XM(0xffffu, 0u), XM(0xfffu, 0u), XM(0xffu, 0u), XM(0x3fu, 0u), XM(0x1fu, 0u),
                        XM(0xfu, 0u), XM(0x3u, 0u), XM(0x1u, 0u), XM(0x00010001u, 0x00010001u), XM(0x00030003u, 0x00030003u), XM(0x000f000fu, 0x000f000fu),
                        XM(0x001f001fu, 0x001f001fu), XM(0x00ff00ffu, 0x00ff00ffu), XM(0x11111111u, 0x11111111u), XM(0x01010101u, 0x01010101u),
                        XM(0x55555555u, 0x55555555u), XM(0xffff0000u, 0u), XM(0u, 0xffffu), XM(0x0000ffffu, 0x0000000fu), XM(0x80000000u, 0x1u)};
        for (int wps : {8})
            for (const X &x : xs) {
                const double ms = time_kernel(x.k, wps, dout), ma = time_kernel(x.ka, wps, dout);
                printf("{\"exec_lo_hi\": \"%s\", \"waves_per_simd\": %d, \"v_mul_u32_u24_ns_per_wave_instr\": %.3f, \"v_add_u32_ns_per_wave_instr\": %.3f}\n", x.name, wps,
                       ms * 1e6 / ((double)wps * ITER * 64.0), ma * 1e6 / ((double)wps * ITER * 64.0));
            }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "sparse")) {
        struct S { const char *name; void (*full)(uint32_t *, uint32_t); void (*one)(uint32_t *, uint32_t); void (*eight)(uint32_t *, uint32_t); int per; };
#define SP(NAME, PER) {#NAME, ks_##NAME<0xffffffffu, 0xffffffffu>, ks_##NAME<1u, 0u>, ks_##NAME<0x01010101u, 0x01010101u>, PER}
        const S ss[] = {SP(add_u32, 1), SP(sub, 1), SP(and, 1), SP(xor, 1), SP(mov, 1), SP(lshlrev, 1), SP(max, 1), SP(mul_u24, 1), SP(mul_lo, 1), SP(mad_u24, 1),
                        SP(lshl_add, 1), SP(add3, 1), SP(or3, 1), SP(cndmask64v, 1), SP(cmp, 1), SP(cvt_ubyte, 1), SP(cvt_f32_u32, 1), SP(ffbh, 1), SP(bfe, 1), SP(perm, 1),
                        SP(fma_f32, 1), SP(add_f32, 1), SP(readlane, 2)};
        for (const S &x : ss) {
            const double n = 8.0 * ITER * 64.0 * x.per;
            printf("{\"instr\": \"%s\", \"waves_per_simd\": 8, \"ns_all_lanes\": %.3f, \"ns_lane0_only\": %.3f, \"ns_8_lanes_strided\": %.3f}\n", x.name,
                   time_kernel(x.full, 8, dout) * 1e6 / n, time_kernel(x.one, 8, dout) * 1e6 / n, time_kernel(x.eight, 8, dout) * 1e6 / n);
        }
        return 0;
    }
    struct T {
        const char *name;
        void (*k)(uint32_t *, uint32_t);
        int div;  // ITER divisor (LDS tests run ITER/4)
        int per;  // instructions per STEP
    };
    const T tests[] = {
        {"v_add_u32", k_add_u32, 1, 1}, {"v_mul_lo_u32", k_mul_lo, 1, 1}, {"v_mul_hi_u32", k_mul_hi, 1, 1},
        {"v_mul_u32_u24", k_mul_u24, 1, 1}, {"v_mad_u32_u24", k_mad_u24, 1, 1}, {"v_lshl_add_u32", k_lshl_add, 1, 1},
        {"v_add3_u32", k_add3, 1, 1}, {"v_cndmask_b32", k_cndmask, 1, 1}, {"v_cndmask_b32_e64 0,1,sgpr", k_cndmask64c, 1, 1},
        {"v_cndmask_b32_e64 v,v,sgpr", k_cndmask64v, 1, 1}, {"v_cmp_lt_u32", k_cmp, 1, 1},
        {"v_cvt_f32_ubyte1", k_cvt_ubyte, 1, 1}, {"v_ffbh_u32", k_ffbh, 1, 1}, {"v_bfe_u32", k_bfe, 1, 1},
        {"v_perm_b32", k_perm, 1, 1}, {"v_mov_b32_dpp", k_mov_dpp, 1, 1}, {"v_add_u32_dpp", k_add_dpp, 1, 1},
        {"v_readlane+v_add", k_readlane, 1, 2}, {"v_fma_f32", k_fma_f32, 1, 1}, {"v_pk_add_u16", k_pk_add_u16, 1, 1},
        {"v_sad_u8", k_sad_u8, 1, 1}, {"s_add+s_lshl", k_salu, 1, 2},
        {"v_lshl_add_u64", k_lshl_add_u64, 1, 1}, {"v_lshlrev_b64", k_lshlrev_b64, 1, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1, 1},
        {"v_add_f64", k_add_f64, 1, 1}, {"v_add_co+v_addc", k_addc_pair, 1, 2},
        {"ds_read_b64", k_ds_read_b64, 4, 1}, {"ds_read_b32", k_ds_read_b32, 4, 1}, {"ds_read_u8", k_ds_read_u8, 4, 1},
        {"ds_read_b64 pitch88", k_ds_read_b64_s88, 4, 1}, {"ds_write_b8", k_ds_write_b8, 4, 1},
        {"ds_write_b64", k_ds_write_b64, 4, 1}, {"ds_bpermute_b32", k_ds_bpermute, 4, 1},
    };
    double base = 0;
    for (int wps : {8, 1}) {
        for (const T &t : tests) {
            const double ms = time_kernel(t.k, wps, dout);
            const double n = (double)wps * (ITER / t.div) * 64.0 * t.per;  // wave-instructions per SIMD
            const double ns = ms * 1e6 / n;
            if (!strcmp(t.name, "v_add_u32")) base = ns;
            printf("{\"test\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"ns_per_wave_instr\": %.4f, \"rel_v_add_u32\": %.3f}\n",
                   t.name, wps, ms, ns, ns / base);
        }
    }
    return 0;
}
