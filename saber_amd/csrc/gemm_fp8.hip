// MXFP8 GEMM on the block-scaled MFMA of gfx950 (BASELINE configs[4]: "Hiera-L fp8 weights on CDNA4 fp8 MFMA").
//
//   C[m][n] = act( sum_b 2^(sa[m][b] + sw[n][b] - 254) * sum_{k in block b} A8[m][k] W8[n][k]  + bias[n] ) (+ res)
//
// Both operands are in the OCP MX format the instruction consumes: e4m3fn elements, K contiguous, one e8m0 scale byte per 32 consecutive
// K-elements (K padded to a multiple of 128: zero elements, scale byte 127).  The scale bytes of operand X live K-step-major,
// SX[K / 128][rows][4]: the dword of one row and one 128-element K-step.  Scales are applied BY the instruction
// (v_mfma_scale_f32_16x16x128_f8f6f4), so the fp32 accumulators need no epilogue scaling, and a block's outliers cost only that block
// its resolution.  Quantisation rule everywhere (weights on the host, activations in ln_mx / quant_mx / the MX epilogue below, the oracle
// in oracle/fp8_ref.py): scale = the smallest power of two with amax <= 448 * scale, elements round-to-nearest-even.
//
// Lane layout of the instruction, probed on the hardware (tools/probes/mx_probe.py): lane (fi = l & 15, fg = l >> 4) supplies row fi of
// its operand; its register bytes 0..15 are K = 16 fg .. 16 fg + 15 and bytes 16..31 are K = 64 + 16 fg .. 64 + 16 fg + 15 of the K-step;
// the scale byte of K block b (K = 32 b .. 32 b + 31) is taken from lane fg = b.  D[row 4 fg + r of the A operand][column fi of the B operand].
//
// Kernel: persistent 256 x 192 tiles (every N of the Hiera-L block GEMMs is a multiple of 192; L2 -> LDS bytes per FLOP are 0.6 of a
// 128 x 128 tile's, the path that bounds this engine's GEMMs: DESIGN.md section 4), 8 waves = 4 (M) x 2 (N), wave = 64 x 96 = 4 x 6 MFMA
// tiles, one K-step = 128 bytes per row = ONE instruction per MFMA tile.  Operands and scale dwords go global -> LDS directly
// (global_load_lds, source-side XOR swizzle of the 16-B chunks so that the ds_read_b128 fragment reads are conflict-free), two 58-KB
// stages, one raw barrier per K-step, ONE continuous K-step stream over the workgroup's tiles (XCD-aware order): the first K-step of the
// next tile is in flight during a tile's epilogue and the epilogue's stores drain under the next tile's products.  Swapped operands
// (W rows = MFMA A operand): a lane owns 4 consecutive output columns of one row.
#include "common.h"
#include "kernels.h"

typedef int mx_v8i __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) const void* mx_gptr;
typedef __attribute__((address_space(3))) void* mx_lptr;

#define MX_TM 256
#define MX_TN 192
#define MX_BK 128                                   // bytes = K-elements per step
#define MX_A_BYTES (MX_TM * MX_BK)                  // 32 KB
#define MX_W_BYTES (MX_TN * MX_BK)                  // 24 KB
#define MX_SA_OFF (MX_A_BYTES + MX_W_BYTES)
#define MX_SW_OFF (MX_SA_OFF + 1024)
#define MX_STAGE (MX_SW_OFF + 1024)                 // 59 392 B
#define MX_LDS (2 * MX_STAGE)
#define MX_BIAS_MAX 11136                          // floats of bias behind the stages (N rounded up to 192): what the 160-KB LDS leaves

enum { MX_OUT_BF16 = 0, MX_OUT_MX8 = 1, MX_OUT_F32 = 2 };

// e8m0 byte of the block scale for a block whose largest magnitude is amax: the smallest e with amax <= 448 * 2^e = 1.75 * 2^(e + 8)
__device__ __forceinline__ int mx_scale_exp(float amax) {
    const uint32_t ab = __float_as_uint(amax);
    const int e = (int)(ab >> 23) - 127 - 8 + ((ab & 0x7fffffu) > 0x600000u ? 1 : 0);
    return max(e, -127);
}
__device__ __forceinline__ float mx_inv_scale(int e) { return __uint_as_float((uint32_t)(127 - e) << 23); }      // 2^-e (e in [-127, 120])
__device__ __forceinline__ uint32_t mx_pack4(float a, float b, float c, float d) {      // |x| <= 448 by construction of the scale
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}

// Epilogue of one wave: acc[j][i][r] = C[m0 + 16 i + fi][n0 + 16 j + 4 fg + r] (i < MI activation tiles, 6 weight tiles = 96 columns = three MX blocks).
// A lane of the accumulator layout owns 4 consecutive columns of ONE row: stored directly, a wave-instruction would write 16 rows x 16-32 B -
// sixteen partial lines through the same address path that carries the LDS-DMA stream (measured: the K = 576 shapes spent as long in the
// epilogue as in the K loop, and the transfer issue of the next tile slowed down under it).  So each 16-row (fp8: 32-row) block is transposed
// through `scr`, >= 4 KB of LDS private to the wave (rows of the stage consumed last that only this wave reads, or any stage after a
// workgroup barrier), and leaves as 16 B per lane, whole 96 / 192-byte row segments per instruction.  LDS traffic is inline asm: a
// compiler-visible ds access behind the direct-to-LDS loads in flight would be ordered with s_waitcnt vmcnt(0).
#define MX_DSW32(addr, v) asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory")
#define MX_DSW64(addr, v) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory")
#define MX_DSW128(addr, v) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory")
#define MX_DSR128(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory")
template <int OUT, int MI>
__device__ __forceinline__ void mx_epilogue(const GemmMxParams& p, f32x4 (&acc)[6][MI], const float4 (&bias4)[6], int64_t m0, int n0, int lane, uint32_t scr) {
    const int fi = lane & 15, fg = lane >> 4;
    if (OUT == MX_OUT_BF16) {
        constexpr int PITCH = 208;                  // 192 B of a row + 16: the 16 rows of a ds_write_b64 spread over the banks
        const int row0 = lane / 12, ch0 = lane - 12 * row0;      // read-back: 16 rows x 12 chunks of 16 B = 3 x 64 lanes
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float v0 = acc[j][i][0] + bias4[j].x, v1 = acc[j][i][1] + bias4[j].y, v2 = acc[j][i][2] + bias4[j].z, v3 = acc[j][i][3] + bias4[j].w;
                if (p.act == ACT_GELU) { const f32x2 g0_ = gelu_erf2((f32x2){v0, v1}), g1_ = gelu_erf2((f32x2){v2, v3}); v0 = g0_.x; v1 = g0_.y; v2 = g1_.x; v3 = g1_.y; }
                const uint64_t pk = ((uint64_t)pack_op16(v2, v3) << 32) | pack_op16(v0, v1);
                MX_DSW64(scr + fi * PITCH + (16 * j + 4 * fg) * 2, pk);
            }
            u32x4 val[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) { const int L = 64 * t + lane, row = L / 12, ch = L - 12 * row; MX_DSR128(val[t], scr + row * PITCH + ch * 16); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]));
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int L = 64 * t + lane, row = L / 12, ch = L - 12 * row;
                const int64_t m = m0 + 16 * i + row;
                const int n = n0 + 8 * ch;
                if (m < p.M && n < p.N) *reinterpret_cast<u32x4*>(p.Cb + m * p.ldcb + n) = val[t];
            }
        }
        (void)row0; (void)ch0;
    } else if (OUT == MX_OUT_F32) {
        constexpr int PITCH = 208;                  // half a block at a time: 16 rows x 48 columns of fp32 = 192 B per row
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float4 r[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {       // residual rows first: they queue behind the K-step transfer in flight
                    const int L = 64 * t + lane, row = L / 12, ch = L - 12 * row;
                    const int64_t m = m0 + 16 * i + row;
                    const int n = n0 + 48 * h + 4 * ch;
                    r[t] = (p.res && m < p.M && n < p.N) ? *reinterpret_cast<const float4*>(p.res + m * p.ldres + n) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    const int j = 3 * h + jj;
                    const u32x4 v = {__float_as_uint(acc[j][i][0] + bias4[j].x), __float_as_uint(acc[j][i][1] + bias4[j].y), __float_as_uint(acc[j][i][2] + bias4[j].z),
                                     __float_as_uint(acc[j][i][3] + bias4[j].w)};
                    MX_DSW128(scr + fi * PITCH + (16 * jj + 4 * fg) * 4, v);
                }
                u32x4 val[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) { const int L = 64 * t + lane, row = L / 12, ch = L - 12 * row; MX_DSR128(val[t], scr + row * PITCH + ch * 16); }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]));
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int L = 64 * t + lane, row = L / 12, ch = L - 12 * row;
                    const int64_t m = m0 + 16 * i + row;
                    const int n = n0 + 48 * h + 4 * ch;
                    const float4 o = make_float4(__uint_as_float(val[t][0]) + r[t].x, __uint_as_float(val[t][1]) + r[t].y, __uint_as_float(val[t][2]) + r[t].z,
                                                 __uint_as_float(val[t][3]) + r[t].w);
                    if (m < p.M && n < p.N) {
                        *reinterpret_cast<float4*>(p.Cf + m * p.ldcf + n) = o;
                        if (p.Cb) *reinterpret_cast<uint2*>(p.Cb + m * p.ldcb + n) = make_uint2(pack_op16(o.x, o.y), pack_op16(o.z, o.w));
                    }
                }
            }
        }
    } else {
        // MX output: a 32-column block of a row = two MFMA tiles x the row's four fg lanes; two 16-row blocks (32 x 96 B) per transposition
        constexpr int PITCH = 112;
#pragma unroll
        for (int i0 = 0; i0 < MI; i0 += 2) {
            constexpr int dummy = 0; (void)dummy;
            const int nrows = (i0 + 1 < MI) ? 32 : 16;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = i0 + ii;
                if (i >= MI) break;
                const int64_t m = m0 + 16 * i + fi;
#pragma unroll
                for (int jp = 0; jp < 3; ++jp) {
                    float v[8];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = 2 * jp + h;
                        v[4 * h + 0] = acc[j][i < MI ? i : 0][0] + bias4[j].x; v[4 * h + 1] = acc[j][i < MI ? i : 0][1] + bias4[j].y;
                        v[4 * h + 2] = acc[j][i < MI ? i : 0][2] + bias4[j].z; v[4 * h + 3] = acc[j][i < MI ? i : 0][3] + bias4[j].w;
                    }
                    if (p.act == ACT_GELU) {
#pragma unroll
                        for (int q = 0; q < 8; q += 2) { const f32x2 g = gelu_erf2((f32x2){v[q], v[q + 1]}); v[q] = g.x; v[q + 1] = g.y; }
                    }
                    float am = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) am = fmaxf(am, fabsf(v[q]));
                    am = xor32_max(xor16_max(am));
                    const int e = mx_scale_exp(am);
                    const float inv = mx_inv_scale(e);
                    const uint32_t q0 = mx_pack4(v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv), q1 = mx_pack4(v[4] * inv, v[5] * inv, v[6] * inv, v[7] * inv);
                    const uint32_t a = scr + (16 * ii + fi) * PITCH + 32 * jp + 4 * fg;
                    MX_DSW32(a, q0);
                    MX_DSW32(a + 16, q1);
                    const int n = n0 + 32 * jp;
                    if (fg == 0 && m < p.M && n < p.N) p.SC[((int64_t)(n >> 7) * p.sc_rows + m) * 4 + ((n >> 5) & 3)] = (uint8_t)(e + 127);
                }
            }
            u32x4 val[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) { const int L = 64 * t + lane, row = L / 6, ch = L - 6 * row; MX_DSR128(val[t], scr + row * PITCH + ch * 16); }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(val[0]), "+v"(val[1]), "+v"(val[2]));
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int L = 64 * t + lane, row = L / 6, ch = L - 6 * row;
                const int64_t m = m0 + 16 * i0 + row;
                const int n = n0 + 16 * ch;
                if (row < nrows && m < p.M && n < p.N) *reinterpret_cast<u32x4*>(p.C8 + m * p.ldc8 + n) = val[t];
            }
        }
    }
}

#define MX_STAMP(k) do { if (STAMPS) { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ts[k] += _n - tprev; tprev = _n; } } while (0)
template <int OUT, bool STAMPS>
__global__ __launch_bounds__(512) void gemm_mx_kernel(GemmMxParams p, unsigned long long* stamps) {
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;     // development (STAMPS): cycles per phase, summed over the workgroup's K-steps
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fi = lane & 15, fg = lane >> 4;
    const int tiles_m = (int)((p.M + MX_TM - 1) / MX_TM), tiles_n = (p.N + MX_TN - 1) / MX_TN;
    const int padded = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const int nk = p.Kp / MX_BK;
    // XCD-aware tile order: linear slots L, L + 8, ... share an XCD's L2 and walk the N tiles of the same M tiles (gridDim.x % 8 == 0)
    auto tile_of = [&](int L, int* tm, int* tn) {
        const int q = L >> 3;
        *tn = q % tiles_n;
        *tm = (q / tiles_n) * 8 + (L & 7);
        return *tm < tiles_m;
    };
    auto next_tile = [&](int L, int* tm, int* tn) {
        while (L < padded && !tile_of(L, tm, tn)) L += gridDim.x;
        return L;
    };
    // ---- issue cursor: 8 direct-to-LDS wave-instructions per wave and K-step: 4 of the 32 activation pieces (1 KB = 8 rows x 128 B), 3 of
    // the 24 weight pieces, and a quarter (64 rows x 4 B) of one of the two scale panels (waves 0-3: activations, 4-6: weights, 7 repeats 6)
    int Li, tmi = 0, tni = 0, kti = 0;
    uint32_t asrc[4], wsrc[3];                      // byte offsets from p.A / p.W (the operands are < 4 GB: checked by the launcher)
    const uint8_t* ssrc;
    int achunk[4], wchunk[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int row = 8 * (4 * wave + i) + (lane >> 3); achunk[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 16; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int row = 8 * (3 * wave + i) + (lane >> 3); wchunk[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 16; }
    const int sq = wave < 4 ? wave : min(wave - 4, 2);
    const int64_t sstride = (wave < 4 ? p.sa_rows : p.sw_rows) * 4;
    const int sdst = (wave < 4 ? MX_SA_OFF : MX_SW_OFF) + sq * 256;
    auto set_issue_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) asrc[i] = (uint32_t)(min((int64_t)tmi * MX_TM + 8 * (4 * wave + i) + (lane >> 3), p.M - 1) * p.lda + achunk[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) wsrc[i] = (uint32_t)((int64_t)min(tni * MX_TN + 8 * (3 * wave + i) + (lane >> 3), p.N - 1) * p.ldw + wchunk[i]);
        ssrc = wave < 4 ? p.SA + ((int64_t)tmi * MX_TM + 64 * sq + lane) * 4 : p.SW + ((int64_t)tni * MX_TN + 64 * sq + lane) * 4;
    };
    auto issue = [&](int stage) {
        char* sx = smem + stage * MX_STAGE;
        const uint32_t ko = (uint32_t)kti * MX_BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((mx_gptr)(p.A + (asrc[i] + ko)), (mx_lptr)(sx + (4 * wave + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) __builtin_amdgcn_global_load_lds((mx_gptr)(p.W + (wsrc[i] + ko)), (mx_lptr)(sx + MX_A_BYTES + (3 * wave + i) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((mx_gptr)(ssrc + kti * sstride), (mx_lptr)(sx + sdst), 4, 0, 0);
        if (++kti == nk) {
            kti = 0;
            Li = next_tile(Li + gridDim.x, &tmi, &tni);
            if (Li < padded) set_issue_tile();
        }
    };
    Li = next_tile(blockIdx.x, &tmi, &tni);
    if (Li >= padded) return;                       // workgroup-uniform
    set_issue_tile();
    int Lc = Li, tmc = tmi, tnc = tni, ktc = 0;     // compute cursor

    // ---- fragment addresses.  Row r of a panel is 128 B; logical chunk c sits at physical chunk c ^ ((r >> 1) & 7); the key is the same for
    // the wave's tiles (16 rows apart).  Lane (fi, fg) reads logical chunks fg and 4 + fg, and the scale byte fg of its row's dword.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(mx_lptr)smem;
    const int key = (fi >> 1) & 7;
    const int ra = wm * 64 + fi, rw = wn * 96 + fi;
    const uint32_t a_lo = lds0 + ra * MX_BK + ((fg ^ key) << 4), a_hi = lds0 + ra * MX_BK + (((4 + fg) ^ key) << 4);
    const uint32_t w_lo = lds0 + MX_A_BYTES + rw * MX_BK + ((fg ^ key) << 4), w_hi = lds0 + MX_A_BYTES + rw * MX_BK + (((4 + fg) ^ key) << 4);
    const uint32_t sa_ad = lds0 + MX_SA_OFF + ra * 4 + fg, sw_ad = lds0 + MX_SW_OFF + rw * 4 + fg;

    f32x4 acc[6][4];            // [weight tile j][activation tile i]
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The bias vector lives in the LDS behind the two stages (this kernel leaves 45 KB free): a global load inside the epilogue queues behind
    // the K-step transfer in flight, and 24 more registers across the K loop make hipcc spill the transfer addresses.
    {
        float* bl = reinterpret_cast<float*>(smem + MX_LDS);
        const int np = tiles_n * MX_TN;
        for (int n = tid; n < np; n += 512) bl[n] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        __syncthreads();
    }

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int stage = 0;
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    while (Lc < padded) {
        __builtin_amdgcn_s_barrier();               // every wave's pieces of `stage` have landed; nobody still reads the other stage
        __builtin_amdgcn_sched_barrier(0);
        MX_STAMP(0);
        if (Li < padded) issue(stage ^ 1);
        MX_STAMP(1);
        // Fragment reads through inline asm: a compiler-visible ds_read behind the direct-to-LDS loads just issued is ordered with
        // s_waitcnt vmcnt(0), which would serialise the next K-step's transfer with this K-step's products.  LDS returns in order: the
        // counted waits below release the products of activation tile i as soon as its two reads are back.
        const uint32_t so = stage * MX_STAGE;
        u32x4 wl[6], wh[6], al[4], ah[4];
        uint32_t sws[6], sas[4];
#define MX_RD8(dst, addr) asm volatile("ds_read_u8 %0, %1" : "=v"(dst) : "v"(addr))
#define MX_RD128(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))
#pragma unroll
        for (int j = 0; j < 6; ++j) MX_RD8(sws[j], sw_ad + so + 64 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i) MX_RD8(sas[i], sa_ad + so + 64 * i);
        // LGKM_CNT is a 4-bit counter that wraps (tools/probes/lgkm_probe.hip: with 16 or 24 reads in flight a counted wait lets values through
        // that have not landed): never more than 15 LDS operations outstanding, so the 30 reads of a K-step go out in three batches, each
        // behind a wait that leaves room for it (the scale bytes and the first fragments have long returned by then)
#pragma unroll
        for (int j = 0; j < 2; ++j) { MX_RD128(wl[j], w_lo + so + 2048 * j); MX_RD128(wh[j], w_hi + so + 2048 * j); }
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
#pragma unroll
        for (int j = 2; j < 6; ++j) { MX_RD128(wl[j], w_lo + so + 2048 * j); MX_RD128(wh[j], w_hi + so + 2048 * j); }
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) { MX_RD128(al[i], a_lo + so + 2048 * i); MX_RD128(ah[i], a_hi + so + 2048 * i); }
        MX_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(6)"
                     : "+v"(sws[0]), "+v"(sws[1]), "+v"(sws[2]), "+v"(sws[3]), "+v"(sws[4]), "+v"(sws[5]), "+v"(sas[0]), "+v"(sas[1]), "+v"(sas[2]), "+v"(sas[3]),
                       "+v"(wl[0]), "+v"(wl[1]), "+v"(wl[2]), "+v"(wl[3]), "+v"(wl[4]), "+v"(wl[5]), "+v"(wh[0]), "+v"(wh[1]), "+v"(wh[2]), "+v"(wh[3]), "+v"(wh[4]),
                       "+v"(wh[5]), "+v"(al[0]), "+v"(ah[0]));
        MX_STAMP(3);
        mx_v8i wf[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] = (mx_v8i){(int)wl[j][0], (int)wl[j][1], (int)wl[j][2], (int)wl[j][3], (int)wh[j][0], (int)wh[j][1], (int)wh[j][2], (int)wh[j][3]};
#define MX_TILE_ROW(i)                                                                                                                                   \
        {                                                                                                                                                \
            const mx_v8i af = (mx_v8i){(int)al[i][0], (int)al[i][1], (int)al[i][2], (int)al[i][3], (int)ah[i][0], (int)ah[i][1], (int)ah[i][2], (int)ah[i][3]}; \
            _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                                                                \
                acc[j][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j], af, acc[j][i], 0, 0, 0, (int)sws[j], 0, (int)sas[i]);                \
        }
        MX_TILE_ROW(0)
        __builtin_amdgcn_sched_barrier(0);          // (keeps each group of products above the next wait)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(al[1]), "+v"(ah[1]));
        MX_TILE_ROW(1)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(al[2]), "+v"(ah[2]));
        MX_TILE_ROW(2)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[3]), "+v"(ah[3]));
        MX_TILE_ROW(3)
        // (the MFMAs touch no memory: without the fence the scheduler hoists the wait for the next K-step's transfer above them)
        __builtin_amdgcn_sched_barrier(0);
        MX_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // next K-step landed (this wave's pieces); the previous tile's stores have drained
        __builtin_amdgcn_sched_barrier(0);
        MX_STAMP(5);
        stage ^= 1;
        if (++ktc < nk) continue;
        // ---------------- epilogue: acc[j][i][r] = C[m0 + 64 wm + 16 i + fi][n0 + 96 wn + 16 j + 4 fg + r]; stores stay in flight into the next tile
        const int64_t m0 = (int64_t)tmc * MX_TM + wm * 64;
        const int n0 = tnc * MX_TN + wn * 96;
        __builtin_amdgcn_s_barrier();               // the stage consumed last becomes the waves' transposition scratch: its readers must be done
        float4 bias4[6];
        {
            u32x4 b[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) MX_DSR128(b[j], lds0 + MX_LDS + (n0 + 16 * j + 4 * fg) * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]));
#pragma unroll
            for (int j = 0; j < 6; ++j) bias4[j] = make_float4(__uint_as_float(b[j][0]), __uint_as_float(b[j][1]), __uint_as_float(b[j][2]), __uint_as_float(b[j][3]));
        }
        mx_epilogue<OUT, 4>(p, acc, bias4, m0, n0, lane, lds0 + (stage ^ 1) * MX_STAGE + wave * 4096);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ktc = 0;
        Lc = next_tile(Lc + gridDim.x, &tmc, &tnc);
        MX_STAMP(6);
    }
    if (STAMPS && lane == 0)
        for (int k = 0; k < 8; ++k) stamps[((int64_t)blockIdx.x * 8 + wave) * 8 + k] = ts[k];
}

// ------------------------------------------------------------------------------------------------ 192 x 96 tiles, two workgroups per CU
// The two kernels above keep eight waves in lockstep (one barrier per K-step), so the phases of a K-step - transfer issue, fragment reads,
// products, and the tile's epilogue (for K = 576 as long as the K loop: 96 GELUs per lane) - run one after the other on the whole CU.  Here
// a workgroup is four waves (one per SIMD) on a 192 x 96 tile with two 38-KB stages, so TWO independent workgroups share a CU and drift
// apart: while one reads fragments, issues its transfer or runs its epilogue, the other's products keep the MFMA pipe busy - the
// two-workgroup form that the engine's bf16 kernel of the same shapes uses (gemm.hip: gemm_bf16_glds2_kernel).  Wave = 48 x 96 = 3 x 6
// MFMA tiles (every wave reads the whole 96-row weight panel).  Per K-step and wave: 6 activation pieces, 3 weight pieces, and the scale
// dwords (waves 0-2: 64 activation rows each; wave 3: the 96 weight rows in two instructions).
#define MX4_TM 192
#define MX4_TN 96
#define MX4_A_BYTES (MX4_TM * MX_BK)                 // 24 KB
#define MX4_W_BYTES (MX4_TN * MX_BK)                 // 12 KB
#define MX4_SA_OFF (MX4_A_BYTES + MX4_W_BYTES)
#define MX4_SW_OFF (MX4_SA_OFF + 1024)
#define MX4_STAGE (MX4_SW_OFF + 512)                 // 38 400 B
#define MX4_LDS (2 * MX4_STAGE)

template <int OUT, bool STAMPS>
__global__ __launch_bounds__(256, 2) void gemm_mx4_kernel(GemmMxParams p, unsigned long long* stamps) {
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, fg = lane >> 4;
    const int tiles_m = (int)((p.M + MX4_TM - 1) / MX4_TM), tiles_n = (p.N + MX4_TN - 1) / MX4_TN;
    const int padded = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const int nk = p.Kp / MX_BK;
    auto tile_of = [&](int L, int* tm, int* tn) {
        const int q = L >> 3;
        *tn = q % tiles_n;
        *tm = (q / tiles_n) * 8 + (L & 7);
        return *tm < tiles_m;
    };
    auto next_tile = [&](int L, int* tm, int* tn) {
        while (L < padded && !tile_of(L, tm, tn)) L += gridDim.x;
        return L;
    };
    // ---- issue cursor
    int Li, tmi = 0, tni = 0, kti = 0;
    uint32_t asrc[6], wsrc[3];                      // byte offsets from p.A / p.W
    const uint8_t* ssrc[2];
    int achunk[6], wchunk[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) { const int row = 8 * (6 * wave + i) + (lane >> 3); achunk[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 16; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int row = 8 * (3 * wave + i) + (lane >> 3); wchunk[i] = ((lane & 7) ^ ((row >> 1) & 7)) * 16; }
    const int64_t sstride = (wave < 3 ? p.sa_rows : p.sw_rows) * 4;
    auto set_issue_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 6; ++i) asrc[i] = (uint32_t)(min((int64_t)tmi * MX4_TM + 8 * (6 * wave + i) + (lane >> 3), p.M - 1) * p.lda + achunk[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) wsrc[i] = (uint32_t)((int64_t)min(tni * MX4_TN + 8 * (3 * wave + i) + (lane >> 3), p.N - 1) * p.ldw + wchunk[i]);
        if (wave < 3) {
            ssrc[0] = p.SA + ((int64_t)tmi * MX4_TM + 64 * wave + lane) * 4;
            ssrc[1] = ssrc[0];
        } else {        // (rows past the panel are clamped: their scale bytes feed no stored output)
            ssrc[0] = p.SW + (int64_t)min((int64_t)tni * MX4_TN + lane, p.sw_rows - 1) * 4;
            ssrc[1] = p.SW + (int64_t)min((int64_t)tni * MX4_TN + 64 + lane, p.sw_rows - 1) * 4;
        }
    };
    auto issue = [&](int stage) {
        char* sx = smem + stage * MX4_STAGE;
        const uint32_t ko = (uint32_t)kti * MX_BK;
#pragma unroll
        for (int i = 0; i < 6; ++i) __builtin_amdgcn_global_load_lds((mx_gptr)(p.A + (asrc[i] + ko)), (mx_lptr)(sx + (6 * wave + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 3; ++i) __builtin_amdgcn_global_load_lds((mx_gptr)(p.W + (wsrc[i] + ko)), (mx_lptr)(sx + MX4_A_BYTES + (3 * wave + i) * 1024), 16, 0, 0);
        if (wave < 3) {
            __builtin_amdgcn_global_load_lds((mx_gptr)(ssrc[0] + kti * sstride), (mx_lptr)(sx + MX4_SA_OFF + wave * 256), 4, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((mx_gptr)(ssrc[0] + kti * sstride), (mx_lptr)(sx + MX4_SW_OFF), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((mx_gptr)(ssrc[1] + kti * sstride), (mx_lptr)(sx + MX4_SW_OFF + 256), 4, 0, 0);
        }
        if (++kti == nk) {
            kti = 0;
            Li = next_tile(Li + gridDim.x, &tmi, &tni);
            if (Li < padded) set_issue_tile();
        }
    };
    Li = next_tile(blockIdx.x, &tmi, &tni);
    if (Li >= padded) return;                       // workgroup-uniform
    set_issue_tile();
    int Lc = Li, tmc = tmi, tnc = tni, ktc = 0;     // compute cursor

    const uint32_t lds0 = (uint32_t)(uintptr_t)(mx_lptr)smem;
    const int key = (fi >> 1) & 7;
    const int ra = wave * 48 + fi, rw = fi;
    const uint32_t a_lo = lds0 + ra * MX_BK + ((fg ^ key) << 4), a_hi = lds0 + ra * MX_BK + (((4 + fg) ^ key) << 4);
    const uint32_t w_lo = lds0 + MX4_A_BYTES + rw * MX_BK + ((fg ^ key) << 4), w_hi = lds0 + MX4_A_BYTES + rw * MX_BK + (((4 + fg) ^ key) << 4);
    const uint32_t sa_ad = lds0 + MX4_SA_OFF + ra * 4 + fg, sw_ad = lds0 + MX4_SW_OFF + rw * 4 + fg;

    f32x4 acc[6][3];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 bias4[6];
    auto load_bias = [&](int tn) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int n = tn * MX4_TN + 16 * j + 4 * fg;
            bias4[j] = (p.bias && n < p.N) ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    load_bias(tnc);

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int stage = 0;
    if (STAMPS) tprev = __builtin_amdgcn_s_memtime();
    while (Lc < padded) {
        __builtin_amdgcn_s_barrier();               // every wave's pieces of `stage` have landed; nobody still reads the other stage
        __builtin_amdgcn_sched_barrier(0);
        MX_STAMP(0);
        if (Li < padded) issue(stage ^ 1);
        MX_STAMP(1);
        const uint32_t so = stage * MX4_STAGE;
        u32x4 wl[6], wh[6], al[3], ah[3];
        uint32_t sws[6], sas[3];
#pragma unroll
        for (int j = 0; j < 6; ++j) MX_RD8(sws[j], sw_ad + so + 64 * j);
#pragma unroll
        for (int i = 0; i < 3; ++i) MX_RD8(sas[i], sa_ad + so + 64 * i);
        // (never more than 15 LDS operations outstanding: see the 4-tile kernel above)
#pragma unroll
        for (int j = 0; j < 3; ++j) { MX_RD128(wl[j], w_lo + so + 2048 * j); MX_RD128(wh[j], w_hi + so + 2048 * j); }
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
#pragma unroll
        for (int j = 3; j < 6; ++j) { MX_RD128(wl[j], w_lo + so + 2048 * j); MX_RD128(wh[j], w_hi + so + 2048 * j); }
        asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 3; ++i) { MX_RD128(al[i], a_lo + so + 2048 * i); MX_RD128(ah[i], a_hi + so + 2048 * i); }
        MX_STAMP(2);
        asm volatile("s_waitcnt lgkmcnt(4)"
                     : "+v"(sws[0]), "+v"(sws[1]), "+v"(sws[2]), "+v"(sws[3]), "+v"(sws[4]), "+v"(sws[5]), "+v"(sas[0]), "+v"(sas[1]), "+v"(sas[2]),
                       "+v"(wl[0]), "+v"(wl[1]), "+v"(wl[2]), "+v"(wl[3]), "+v"(wl[4]), "+v"(wl[5]), "+v"(wh[0]), "+v"(wh[1]), "+v"(wh[2]), "+v"(wh[3]), "+v"(wh[4]),
                       "+v"(wh[5]), "+v"(al[0]), "+v"(ah[0]));
        MX_STAMP(3);
        mx_v8i wf[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wf[j] = (mx_v8i){(int)wl[j][0], (int)wl[j][1], (int)wl[j][2], (int)wl[j][3], (int)wh[j][0], (int)wh[j][1], (int)wh[j][2], (int)wh[j][3]};
        MX_TILE_ROW(0)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(al[1]), "+v"(ah[1]));
        MX_TILE_ROW(1)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(al[2]), "+v"(ah[2]));
        MX_TILE_ROW(2)
        __builtin_amdgcn_sched_barrier(0);
        MX_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // next K-step landed (this wave's pieces); the previous tile's stores have drained
        __builtin_amdgcn_sched_barrier(0);
        MX_STAMP(5);
        stage ^= 1;
        if (++ktc < nk) continue;
        // transposition scratch: this wave's own 48 activation rows (6 KB) of the stage consumed last - nobody else reads them
        mx_epilogue<OUT, 3>(p, acc, bias4, (int64_t)tmc * MX4_TM + wave * 48, tnc * MX4_TN, lane, lds0 + (stage ^ 1) * MX4_STAGE + wave * 6144);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        ktc = 0;
        Lc = next_tile(Lc + gridDim.x, &tmc, &tnc);
        if (Lc < padded) load_bias(tnc);
        MX_STAMP(6);
    }
    if (STAMPS && lane == 0)
        for (int k = 0; k < 8; ++k) stamps[((int64_t)blockIdx.x * 4 + wave) * 8 + k] = ts[k];
}

const char* launch_gemm_mx(const GemmMxParams& p, hipStream_t s) {
    if (p.M <= 0 || p.N <= 0) return nullptr;
    if (!p.A || !p.W || !p.SA || !p.SW || p.Kp <= 0 || (p.Kp % MX_BK) || (p.N & 15) || (p.lda & 15) || (p.ldw & 15) || p.lda < p.Kp || p.ldw < p.Kp)
        return "gemm_mx: bad argument (K padded to a multiple of 128, N % 16 == 0, 16-byte aligned rows)";
    // two kernels: 256 x 192 tiles (one 8-wave workgroup per CU: fewest L2 -> LDS bytes per FLOP) for the long-M shapes of stage 2, 192 x 96
    // tiles (two 4-wave workgroups per CU) for the rest; SABER_AMD_MX_KERNEL = 2 | 4 forces one (development A/B)
    static const int forced = getenv("SABER_AMD_MX_KERNEL") ? atoi(getenv("SABER_AMD_MX_KERNEL")) : 0;
    const int ver = forced ? forced : ((p.M >= 40000 && p.N <= MX_BIAS_MAX - 192) ? 2 : 4);
    const int TM = ver == 2 ? MX_TM : MX4_TM, TN = ver == 4 ? MX4_TN : MX_TN;
    const int tiles_m = (int)((p.M + TM - 1) / TM), tiles_n = (p.N + TN - 1) / TN;
    if (p.sa_rows < (p.M + 767) / 768 * 768 || p.sw_rows < (int64_t)(p.N + 191) / 192 * 192)
        return "gemm_mx: scale panels must cover whole tiles (sa_rows >= ceil(M / 768) * 768, sw_rows >= ceil(N / 192) * 192)";
    const int outs = (p.Cf ? 1 : 0) + (p.C8 ? 1 : 0) + ((p.Cb && !p.Cf) ? 1 : 0);
    if (outs != 1) return "gemm_mx: exactly one of Cf (+ optional Cb copy) / Cb / C8";
    if (p.Cf && ((p.ldcf & 3) || (p.res && (p.ldres & 3)) || p.act != ACT_NONE)) return "gemm_mx: fp32 output: leading dimensions % 4 == 0, no activation";
    if (p.Cb && (p.ldcb & 7)) return "gemm_mx: bf16 leading dimension must be a multiple of 8";
    if (p.C8 && (!p.SC || (p.N & 127) || (p.ldc8 & 15) || p.ldc8 < p.N || p.sc_rows < p.M)) return "gemm_mx: MX output needs N % 128 == 0, a scale panel of >= M rows and 16-byte aligned rows";
    if (p.act != ACT_NONE && p.act != ACT_GELU) return "gemm_mx: activation must be none or GELU";
    if (p.M * p.lda >= ((int64_t)1 << 32) || (int64_t)p.N * p.ldw >= ((int64_t)1 << 32)) return "gemm_mx: operand larger than 4 GB";
    static bool attr = false;
    if (!attr) {
        hipError_t st = hipSuccess;
        for (const void* f : {reinterpret_cast<const void*>(gemm_mx_kernel<MX_OUT_BF16, false>), reinterpret_cast<const void*>(gemm_mx_kernel<MX_OUT_MX8, false>),
                              reinterpret_cast<const void*>(gemm_mx_kernel<MX_OUT_F32, false>), reinterpret_cast<const void*>(gemm_mx_kernel<MX_OUT_BF16, true>),
                              reinterpret_cast<const void*>(gemm_mx_kernel<MX_OUT_MX8, true>), reinterpret_cast<const void*>(gemm_mx_kernel<MX_OUT_F32, true>)})
            if (st == hipSuccess) st = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS + MX_BIAS_MAX * 4);
        for (const void* f : {reinterpret_cast<const void*>(gemm_mx4_kernel<MX_OUT_BF16, false>), reinterpret_cast<const void*>(gemm_mx4_kernel<MX_OUT_MX8, false>),
                              reinterpret_cast<const void*>(gemm_mx4_kernel<MX_OUT_F32, false>), reinterpret_cast<const void*>(gemm_mx4_kernel<MX_OUT_BF16, true>),
                              reinterpret_cast<const void*>(gemm_mx4_kernel<MX_OUT_MX8, true>), reinterpret_cast<const void*>(gemm_mx4_kernel<MX_OUT_F32, true>)})
            if (st == hipSuccess) st = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MX4_LDS);
        if (st != hipSuccess) return "gemm_mx: cannot reserve LDS";
        attr = true;
    }
    const int padded = ((tiles_m + 7) / 8) * 8 * tiles_n;
    const int cap = ver == 4 ? 512 : 256;                  // persistent: as many workgroups as fit the chip at once (a multiple of 8)
    const int grid = padded < cap ? padded : cap;
    unsigned long long* sb = g_saber_stamp_buf;
#define MX_LAUNCH(K, LDS, ST, NT)                                                                                               \
    do {                                                                                                                        \
        if (p.Cf) hipLaunchKernelGGL((K<MX_OUT_F32, ST>), dim3(grid), dim3(NT), LDS, s, p, sb);                                  \
        else if (p.C8) hipLaunchKernelGGL((K<MX_OUT_MX8, ST>), dim3(grid), dim3(NT), LDS, s, p, sb);                             \
        else hipLaunchKernelGGL((K<MX_OUT_BF16, ST>), dim3(grid), dim3(NT), LDS, s, p, sb);                                      \
    } while (0)
    if (ver == 2) { const int lds = MX_LDS + tiles_n * MX_TN * 4; if (sb) MX_LAUNCH(gemm_mx_kernel, lds, true, 512); else MX_LAUNCH(gemm_mx_kernel, lds, false, 512); }
    else { if (sb) MX_LAUNCH(gemm_mx4_kernel, MX4_LDS, true, 256); else MX_LAUNCH(gemm_mx4_kernel, MX4_LDS, false, 256); }
    return nullptr;
}

// ------------------------------------------------------------------------------------------------ activation quantisers
// One wave per row; lane l of chunk c owns elements 512 c + 8 l .. + 7, so a 32-element MX block is a quad of lanes.
#define MXQ_CHUNKS 5          // rows up to 2 560 elements stay in registers

template <bool LN>
__global__ __launch_bounds__(256) void quant_mx_kernel(const float* __restrict__ xf, const bf16_t* __restrict__ xb, int64_t ldx, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int C, uint8_t* __restrict__ out, int64_t ldo, int Kp,
                                                       uint8_t* __restrict__ SC, int64_t sc_rows, int64_t M) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[MXQ_CHUNKS][8];
#pragma unroll
    for (int c = 0; c < MXQ_CHUNKS; ++c) {
        const int k = 512 * c + 8 * lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) v[c][q] = 0.f;
        if (k < C) {
            if (xf) {
                const float4 a = *reinterpret_cast<const float4*>(xf + row * ldx + k), b = *reinterpret_cast<const float4*>(xf + row * ldx + k + 4);
                v[c][0] = a.x; v[c][1] = a.y; v[c][2] = a.z; v[c][3] = a.w; v[c][4] = b.x; v[c][5] = b.y; v[c][6] = b.z; v[c][7] = b.w;
            } else {
                const u32x4 u = *reinterpret_cast<const u32x4*>(xb + row * ldx + k);
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[c][2 * q] = __uint_as_float(u[q] << 16); v[c][2 * q + 1] = __uint_as_float(u[q] & 0xffff0000u); }
            }
        }
    }
    if (LN) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MXQ_CHUNKS; ++c)
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[c][q];
        const float mean = wave_sum(s) / (float)C;
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < MXQ_CHUNKS; ++c) {
            if (512 * c + 8 * lane < C) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { const float d = v[c][q] - mean; ss += d * d; }
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
#pragma unroll
        for (int c = 0; c < MXQ_CHUNKS; ++c) {
            const int k = 512 * c + 8 * lane;
            if (k < C) {
                const float4 g0 = *reinterpret_cast<const float4*>(gamma + k), g1 = *reinterpret_cast<const float4*>(gamma + k + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(beta + k), b1 = *reinterpret_cast<const float4*>(beta + k + 4);
                const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) v[c][q] = (v[c][q] - mean) * rstd * g[q] + b[q];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MXQ_CHUNKS; ++c) {
        const int k = 512 * c + 8 * lane;
        if (k >= Kp) continue;                       // (whole quads: Kp % 32 == 0)
        float am = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) am = fmaxf(am, fabsf(v[c][q]));
        am = fmaxf(am, dpp_mov<DPP_XOR1>(am));
        am = fmaxf(am, dpp_mov<DPP_XOR2>(am));
        const int e = k < C ? mx_scale_exp(am) : 0;  // K padding: zeros with a unit scale
        const float inv = mx_inv_scale(e);
        *reinterpret_cast<uint2*>(out + row * ldo + k) =
            make_uint2(mx_pack4(v[c][0] * inv, v[c][1] * inv, v[c][2] * inv, v[c][3] * inv), mx_pack4(v[c][4] * inv, v[c][5] * inv, v[c][6] * inv, v[c][7] * inv));
        const int blk = k >> 5;
        if ((lane & 3) == 0) SC[((int64_t)(blk >> 2) * sc_rows + row) * 4 + (blk & 3)] = (uint8_t)(e + 127);
    }
}

static const char* quant_mx_check(int C, int Kp, int64_t ldx, int64_t ldo, int64_t sc_rows, int64_t M, const void* out, const void* SC) {
    if (!out || !SC || (C & 31) || (Kp & 127) || Kp < C || Kp > 512 * MXQ_CHUNKS || (ldx & 7) || (ldo & 15) || ldo < Kp || sc_rows < M)
        return "quant_mx: C % 32 == 0, Kp a multiple of 128 with C <= Kp <= 2560, 16-byte aligned rows, scale panel of >= M rows";
    return nullptr;
}
// bf16 [M][C] -> MX ([M][Kp] e4m3 + SC[Kp / 128][sc_rows][4] e8m0)
const char* launch_quant_mx_bf16(const bf16_t* x, int64_t ldx, int C, uint8_t* out, int64_t ldo, int Kp, uint8_t* SC, int64_t sc_rows, int64_t M, hipStream_t s) {
    if (M <= 0) return nullptr;
    if (const char* m = quant_mx_check(C, Kp, ldx, ldo, sc_rows, M, out, SC)) return m;
    hipLaunchKernelGGL(quant_mx_kernel<false>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, nullptr, x, ldx, nullptr, nullptr, 0.f, C, out, ldo, Kp, SC, sc_rows, M);
    return nullptr;
}
// LayerNorm over C of fp32 rows, result straight into MX (what the fp8 GEMMs of the encoder blocks consume)
const char* launch_ln_mx(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int C, uint8_t* out, int64_t ldo, int Kp, uint8_t* SC, int64_t sc_rows,
                         int64_t M, hipStream_t s) {
    if (M <= 0) return nullptr;
    if (!x || !gamma || !beta) return "ln_mx: null argument";
    if (const char* m = quant_mx_check(C, Kp, ldx, ldo, sc_rows, M, out, SC)) return m;
    hipLaunchKernelGGL(quant_mx_kernel<true>, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, nullptr, ldx, gamma, beta, eps, C, out, ldo, Kp, SC, sc_rows, M);
    return nullptr;
}
