// cm_linear_bwd.hip - streaming backward of one dense layer for widths 32 / 64 / 128 (every hidden layer of the nets):
//
//   dz = (dy + dy2) * act'(y);  dx = dz.W;  dW += dz^T.x (or x^T.dz);  db += colsum(dz)      reads dy, y, x once; writes dx once
//
// (reference: the autograd of nn.Linear + tanh in garage/torch/modules/multi_headed_mlp_module.py:134-149,
// GraphConvolutionModule's H.W graph_conv_module.py:63, AttentionModule.linear_in attention_module.py:36.)
//
// The first version of this kernel (lin::bwd_kernel in cm_linear.hip, kept for the ragged widths: observation, logits)
// staged a chunk through registers, waited, computed, waited: two workgroups per CU, ~1.3 TB/s.  This one is built
// around the load path:
//   * one persistent 8-wave workgroup per CU walks 64-row chunks; chunk t+1 arrives by LDS-DMA
//     (global_load_lds_dwordx4, no registers) into the second buffer while chunk t is on the matrix pipe;
//   * an LDS-DMA writes 1 KiB contiguously, so the tiles are unpadded rows; bank conflicts of the row-strided operand
//     reads are removed by XOR-swizzling the 16-byte chunk index inside a row - applied to the per-lane SOURCE address
//     on the way in and to the chunk index on every read;
//   * the weight gradient is split by ROWS across waves (wave w: rows 16 (w/2) .. +15, half of the B tiles), over
//     column-permuted tiles (tile u of a 64-wide operand = columns 4c + u), so that ONE ds_read_b128 per operand feeds
//     the four tiles of a k-step; partial sums live in accumulators for the life of the workgroup and are reduced
//     through LDS, then one float atomic per element per workgroup;
//   * dx is computed transposed (weights as the A operand) so that a lane ends with four consecutive input features:
//     one 16-byte store per tile.
#include <stdlib.h>

#include <algorithm>
#include <cmath>

#include "cm_internal.h"

namespace cm {
namespace lin2 {

constexpr int TPB = 512, NW = 8, ROWS = 64;
typedef float v4f __attribute__((ext_vector_type(4)));

// swizzle of the 16-byte chunk index inside row r of a W-wide tile (W / 4 chunks per row; rows of 32 floats cover
// half of the 64 banks, so there the row's parity picks the half and the remaining bits do the swizzle)
template <int W>
__device__ __forceinline__ int swz(int r) { return W == 32 ? ((r >> 1) & 7) : (r & 15); }

// float offset of 16-byte chunk j of row r
template <int W>
__device__ __forceinline__ int chunk_at(int r, int j) { return (r * (W / 4) + (j ^ swz<W>(r))) * 4; }

// issue this wave's LDS-DMA loads of a [ROWS x W] tile to LDS byte address lds_addr (rows past `rows` re-read the last valid row; the caller zeroes them)
template <int W>
__device__ __forceinline__ void issue_tile(unsigned lds_addr, const float *__restrict__ src, long r0, int rows, int wave, int lane) {
    constexpr int CH = W / 4, NI = W / 32;                 // chunks per row; wave-instructions per wave
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int inst = wave * NI + i, p = inst * 64 + lane;
        const int r = p / CH, slot = p % CH;
        const int rr = r < rows ? r : rows - 1;
        const float *g = src + (r0 + rr) * W + 4 * (slot ^ swz<W>(r));
        // hipcc orders every later ds_read behind a builtin LDS-DMA (s_waitcnt vmcnt(0) before the first operand read of
        // the chunk being computed - no overlap left); as an asm statement the DMA is outside its bookkeeping and the
        // kernel waits for it explicitly, before the barrier that hands the buffer over
        const unsigned dst_b = __builtin_amdgcn_readfirstlane(lds_addr + (unsigned)(inst * 1024));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst_b) : "memory");
    }
}

// First-layer form: the source rows are KR < W floats wide (the observation: 21 / 29 / 53 / 77) and only 4-byte aligned.
// One dword per lane (256 bytes per wave-instruction); tile columns >= KR are fetched from a zero word.
__device__ float zero_word[4] = { 0.0f, 0.0f, 0.0f, 0.0f };

template <int W>
__device__ __forceinline__ void issue_tile_ragged(unsigned lds_addr, const float *__restrict__ src, long r0, int rows, int KR, int wave,
                                                  int lane) {
    constexpr int NI = W / 8;                              // ROWS * W dwords / 64 lanes / 8 waves
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int inst = wave * NI + i, p = inst * 64 + lane;
        const int r = p / W, cw = p % W, slot = cw >> 2, w = cw & 3;
        const int rr = r < rows ? r : rows - 1;
        const int col = 4 * (slot ^ swz<W>(r)) + w;
        const float *g = col < KR ? src + (r0 + rr) * KR + col : zero_word;
        const unsigned dst_b = __builtin_amdgcn_readfirstlane(lds_addr + (unsigned)(inst * 256));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst_b) : "memory");
    }
}

// column of element c of permuted tile u of a W-wide operand
template <int W>
__device__ __forceinline__ int col_of(int u, int c) { return W == 32 ? 2 * c + u : 64 * (u >> 2) + 4 * c + (u & 3); }

// the NU values of row r that lane column c feeds to tiles u0 .. u0 + NU - 1
template <int W, int NU>
__device__ __forceinline__ void load_cols(const float *tile, int r, int c, int u0, float (&out)[NU]) {
    if constexpr (W == 128) {
#pragma unroll
        for (int h = 0; h < NU / 4; ++h) {
            const float4 v = *reinterpret_cast<const float4 *>(tile + chunk_at<W>(r, 16 * ((u0 >> 2) + h) + c));
            out[4 * h + 0] = v.x; out[4 * h + 1] = v.y; out[4 * h + 2] = v.z; out[4 * h + 3] = v.w;
        }
    } else if constexpr (W == 64) {
        if constexpr (NU == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(tile + chunk_at<W>(r, c));
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        } else {
            const float2 v = *reinterpret_cast<const float2 *>(tile + chunk_at<W>(r, c) + u0);
            out[0] = v.x; out[1] = v.y;
        }
    } else {
        if constexpr (NU == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(tile + chunk_at<W>(r, c >> 1) + 2 * (c & 1));
            out[0] = v.x; out[1] = v.y;
        } else {
            out[0] = tile[chunk_at<W>(r, c >> 1) + 2 * (c & 1) + u0];
        }
    }
}

// RAG: first-layer form - x rows are KR <= K floats wide (zero-padded to K in LDS), no input gradient, dW is [O][KR]
// WO > 0: two chained layers (the encoder: obs -> x = tanh(.) -> y = tanh(.)).  The input gradient of this layer never
// leaves the workgroup: multiplied by tanh'(x) it overwrites the x tile IN PLACE (every position is produced by exactly one
// lane) and feeds the FIRST layer's weight / bias gradient against a WO-wide observation tile (rows KO <= WO floats wide):
// DW1 [K][KO], DB1 [K]; DX is not written.
template <int KT, int OT, int ACT, int LAYOUT, bool RAG = false, int WO = 0>
__global__ __launch_bounds__(TPB) void bwd_kernel(long R, int KR, const float *__restrict__ X, const float *__restrict__ W,
                                                  const float *__restrict__ DY, const float *__restrict__ DY2,
                                                  const float *__restrict__ Yv,
                                                  float *__restrict__ DX, float *__restrict__ DW, float *__restrict__ DB,
                                                  const float *__restrict__ OBS, int KO, float *__restrict__ DW1,
                                                  float *__restrict__ DB1) {
    constexpr int K = 16 * KT, O = 16 * OT;
    constexpr int ZF = ROWS * O, XF = ROWS * K, OF = ROWS * WO, BUF = ZF + XF + OF;
    static_assert(WO == 0 || (!RAG && LAYOUT == 0), "the chained form is the nn.Linear layout with full-width x rows");
    constexpr int WA = LAYOUT == 0 ? O : K, WB = LAYOUT == 0 ? K : O;   // dW is [WA][WB]
    constexpr int NA = WA / 16, NBH = WB / 32;                          // A tiles; B tiles of this wave's half
    constexpr int NKT = KT / 2;                                         // dx column tiles per wave
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Ys = lds + 2 * BUF;                            // y tile (ACT) and, behind it, the second gradient's tile (DY2)
    float *D2s = Ys + (ACT ? ZF : 0);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int rg = wave >> 1, hb = wave & 1;

    v4f acc[NA][NBH];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NBH; ++j) acc[i][j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
    float zsum[LAYOUT == 0 ? NA : NBH];
#pragma unroll
    for (float &z : zsum) z = 0.0f;

    // dx: row tile rg, input-feature tiles kt = hb * NKT + i; wf[i][4 oq + u] = W(k = 16 kt + c, o = 16 oq + 4 g + u)
    constexpr bool TSPLIT = RAG && OT == 8 && WO == 0;
    v4f accT[TSPLIT ? KT : 1];
#pragma unroll
    for (v4f &t : accT) t = (v4f){ 0.f, 0.f, 0.f, 0.f };
    v4f acc1[4];                                          // chained form: this wave's four dW tiles (see the weight-gradient step)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc1[j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
    float zs1 = 0.0f;
    constexpr int NOB = WO > 0 ? WO / 16 : 1;            // chained form: this wave's dW1 tiles = x columns 16 wave .. +15 x every obs tile
    v4f acc2[NOB];
#pragma unroll
    for (int j = 0; j < NOB; ++j) acc2[j] = (v4f){ 0.f, 0.f, 0.f, 0.f };
    float dbsum1 = 0.0f;
    float wf[RAG ? 1 : NKT][RAG ? 1 : 4 * OT];
    if constexpr (!RAG) if (DX || WO > 0) {
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            const int k = 16 * (hb * NKT + i) + c;
#pragma unroll
            for (int e = 0; e < 4 * OT; ++e) {
                const int o = 16 * (e >> 2) + 4 * g + (e & 3);
                wf[i][e] = LAYOUT == 0 ? W[(size_t)o * K + k] : W[(size_t)k * O + o];
            }
        }
    }

    const long n_chunks = (R + ROWS - 1) / ROWS;
    long ch = blockIdx.x;
    int cb = 0;
    // dz = dy * (1 - y^2) in place, and zero rows past the end of the last chunk (the DMA re-read a valid row there)
    auto finish_tile = [&](float *buf, int rows) {
        float *Zs = buf, *Xs = buf + ZF;
        if (ACT || DY2) {
#pragma unroll
            for (int i = 0; i < ZF / (4 * TPB); ++i) {
                const int p = 4 * (tid + i * TPB);
                float4 z = *reinterpret_cast<float4 *>(Zs + p);
                if (DY2) {
                    const float4 u = *reinterpret_cast<const float4 *>(D2s + p);
                    z.x += u.x; z.y += u.y; z.z += u.z; z.w += u.w;
                }
                if (ACT) {
                    const float4 y = *reinterpret_cast<const float4 *>(Ys + p);
                    z.x *= 1.0f - y.x * y.x; z.y *= 1.0f - y.y * y.y; z.z *= 1.0f - y.z * y.z; z.w *= 1.0f - y.w * y.w;
                }
                if (p >= rows * O) z = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4 *>(Zs + p) = z;
            }
        } else if (rows < ROWS) {
            for (int p = rows * O + tid; p < ZF; p += TPB) Zs[p] = 0.0f;
        }
        if (rows < ROWS) {
            for (int p = rows * K + tid; p < XF; p += TPB) Xs[p] = 0.0f;
            if constexpr (WO > 0) for (int p = rows * WO + tid; p < OF; p += TPB) Xs[XF + p] = 0.0f;
        }
    };
    auto issue = [&](float *buf, long chunk) {
        const long r0 = chunk * ROWS;
        const int rows = (int)min((long)ROWS, R - r0);
        const unsigned b = lds_base + (unsigned)((buf - lds) * sizeof(float));
        issue_tile<O>(b, DY, r0, rows, wave, lane);
        if constexpr (RAG) issue_tile_ragged<K>(b + ZF * (unsigned)sizeof(float), X, r0, rows, KR, wave, lane);
        else issue_tile<K>(b + ZF * (unsigned)sizeof(float), X, r0, rows, wave, lane);
        if constexpr (WO > 0) issue_tile_ragged<WO>(b + (ZF + XF) * (unsigned)sizeof(float), OBS, r0, rows, KO, wave, lane);
        if (ACT) issue_tile<O>(lds_base + 2 * BUF * (unsigned)sizeof(float), Yv, r0, rows, wave, lane);
        if (DY2) issue_tile<O>(lds_base + (2 * BUF + (ACT ? ZF : 0)) * (unsigned)sizeof(float), DY2, r0, rows, wave, lane);
    };
    if (ch < n_chunks) {
        issue(lds, ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int rows = (int)min((long)ROWS, R - ch * ROWS);
        finish_tile(lds, rows);
        __syncthreads();
    }
    for (; ch < n_chunks; ch += gridDim.x, cb ^= 1) {
        const long next = ch + gridDim.x;
        float *cur = lds + cb * BUF, *nxt = lds + (cb ^ 1) * BUF;
        if (next < n_chunks) issue(nxt, next);
        const float *Zs = cur, *Xs = cur + ZF;
        const float *As = LAYOUT == 0 ? Zs : Xs, *Bs = LAYOUT == 0 ? Xs : Zs;
        const long r0 = ch * ROWS;
        const int rows = (int)min((long)ROWS, R - r0);
        // ---- weight gradient ----
        if constexpr (WO > 0) {
            // chained form: split by TILES (wave w: z-column tile w / 2, four x-column tiles of half w % 2, all 64 rows) -
            // 16 accumulator registers instead of 64, complete sums per wave (no cross-wave reduction at the end)
            static_assert(K == 128 && O == 64, "chained form: the 128 -> 64 encoder layer");
            const int ua = wave >> 1, ubh = wave & 1;
#pragma unroll 4
            for (int kk = 0; kk < ROWS / 4; ++kk) {
                const int rr = 4 * kk + g;
                const float a = Zs[chunk_at<O>(rr, c) + ua];                     // z[rr][4 c + ua]: permuted tile ua
                const float4 b = *reinterpret_cast<const float4 *>(Xs + chunk_at<K>(rr, 16 * ubh + c));
                acc1[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.x, acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.y, acc1[1], 0, 0, 0);
                acc1[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.z, acc1[2], 0, 0, 0);
                acc1[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b.w, acc1[3], 0, 0, 0);
                if (ubh == 0) zs1 += a;
            }
        } else if constexpr (TSPLIT) {
            // first layer, 128 outputs: split by tiles as well (wave w: z-column tile w, every x-column tile, all 64 rows):
            // 4 KT accumulator registers instead of 16 KT - the row-split build of the 128-wide case spilled, and a scratch
            // reload waits on vmcnt(0), i.e. on the LDS-DMA queue
#pragma unroll 4
            for (int kk = 0; kk < ROWS / 4; ++kk) {
                const int rr = 4 * kk + g;
                const int zc = col_of<O>(wave, c);
                const float a = Zs[chunk_at<O>(rr, zc >> 2) + (zc & 3)];
                float b[KT];
                load_cols<K, KT>(Xs, rr, c, 0, b);
#pragma unroll
                for (int j = 0; j < KT; ++j) accT[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], accT[j], 0, 0, 0);
                zs1 += a;
            }
        } else
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = 16 * rg + 4 * s + g;
            float a[NA], b[NBH];
            load_cols<WA, NA>(As, r, c, 0, a);
            load_cols<WB, NBH>(Bs, r, c, hb * NBH, b);
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int j = 0; j < NBH; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
            if (DB) {
                if constexpr (LAYOUT == 0) {
                    if (hb == 0) {
#pragma unroll
                        for (int i = 0; i < NA; ++i) zsum[i] += a[i];
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NBH; ++j) zsum[j] += b[j];
                }
            }
        }
        // ---- input gradient, transposed: D[k][row] = sum_o W(k, o) dz[row][o] ----
        if constexpr (!RAG) if (DX || WO > 0) {
            v4f d[NKT];
#pragma unroll
            for (int i = 0; i < NKT; ++i) d[i] = (v4f){ 0.f, 0.f, 0.f, 0.f };
            const int r = 16 * rg + c;
#pragma unroll
            for (int oq = 0; oq < OT; ++oq) {
                const float4 z = *reinterpret_cast<const float4 *>(Zs + chunk_at<O>(r, 4 * oq + g));
#pragma unroll
                for (int i = 0; i < NKT; ++i) {
                    d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][4 * oq + 0], z.x, d[i], 0, 0, 0);
                    d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][4 * oq + 1], z.y, d[i], 0, 0, 0);
                    d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][4 * oq + 2], z.z, d[i], 0, 0, 0);
                    d[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i][4 * oq + 3], z.w, d[i], 0, 0, 0);
                }
            }
            if constexpr (WO == 0) {
                if (r < rows) {
#pragma unroll
                    for (int i = 0; i < NKT; ++i)
                        *reinterpret_cast<float4 *>(DX + (size_t)(r0 + r) * K + 16 * (hb * NKT + i) + 4 * g) =
                            make_float4(d[i][0], d[i][1], d[i][2], d[i][3]);
                }
            } else {
                float *Xw = cur + ZF;
                __syncthreads();                           // every wave is done reading the x tile (weight gradient above)
#pragma unroll
                for (int i = 0; i < NKT; ++i) {           // dz1 = dx * (1 - x^2), in place
                    float4 *px = reinterpret_cast<float4 *>(Xw + chunk_at<K>(r, 4 * (hb * NKT + i) + g));
                    const float4 x = *px;
                    *px = make_float4(d[i][0] * (1.0f - x.x * x.x), d[i][1] * (1.0f - x.y * x.y), d[i][2] * (1.0f - x.z * x.z),
                                      d[i][3] * (1.0f - x.w * x.w));
                }
                __syncthreads();
                // first layer's weight gradient, one x-column tile per wave over all 64 rows: C[p = x col][q = obs col]
                const float *Os = cur + ZF + XF;
#pragma unroll 4
                for (int kk = 0; kk < ROWS / 4; ++kk) {
                    const int rr = 4 * kk + g, col = 16 * wave + c;
                    const float a = Xw[chunk_at<K>(rr, col >> 2) + (col & 3)];
                    dbsum1 += a;
#pragma unroll
                    for (int j = 0; j < NOB; ++j) {
                        const int oc = 16 * j + c;
                        const float b = Os[chunk_at<WO>(rr, oc >> 2) + (oc & 3)];
                        acc2[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc2[j], 0, 0, 0);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // chunk `next` has landed; every wave is done with `cur`
        if (next < n_chunks) {
            const int nrows = (int)min((long)ROWS, R - next * ROWS);
            if (ACT || DY2 || nrows < ROWS) {
                finish_tile(nxt, nrows);
                __syncthreads();
            }
        }
    }

    if constexpr (WO > 0) {
        // the chained layer's gradients are complete per wave (all rows of all its chunks): straight to HBM
#pragma unroll
        for (int j = 0; j < NOB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = 16 * j + c;
                if (q < KO) atomicAdd(DW1 + (size_t)(16 * wave + 4 * g + r) * KO + q, acc2[j][r]);
            }
        if (DB1) {
            float v = dbsum1;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) atomicAdd(DB1 + 16 * wave + c, v);
        }
        // this layer: C[p = z column 4 (4 g + r) + ua][q = x column 64 ubh + 4 c + j]
        const int ua = wave >> 1, ubh = wave & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(DW + (size_t)(4 * (4 * g + r) + ua) * K + 64 * ubh + 4 * c + j, acc1[j][r]);
        if (DB && ubh == 0) {
            float v = zs1;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) atomicAdd(DB + 4 * c + ua, v);
        }
        return;
    }
    if constexpr (TSPLIT) {
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = col_of<K>(j, c);
                if (q < KR) atomicAdd(DW + (size_t)col_of<O>(wave, 4 * g + r) * KR + q, accT[j][r]);
            }
        if (DB) {
            float v = zs1;
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) atomicAdd(DB + col_of<O>(wave, c), v);
        }
        return;
    }
    // ---- reduce the weight-gradient partial sums of the four row groups through LDS; one atomic per element ----
    // lds as [rg][tile][lane] float4; tiles of this half-wave pair: t = i * NBH + j  ->  (ua = i, ub = hb * NBH + j)
    constexpr int NT = NA * NBH;                          // tiles per wave (<= 16)
    float4 *red = reinterpret_cast<float4 *>(lds);        // 4 rg x 2 hb x NT x 64 float4 <= 128 KiB ... done in passes of TP tiles
    constexpr int TP = NT < 4 ? NT : 4;                   // 8 waves x 4 tiles x 1 KiB = 32 KiB per pass
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += TP) {
        __syncthreads();
#pragma unroll
        for (int tl = 0; tl < TP; ++tl) {
            const int t = t0 + tl, i = t / NBH, j = t % NBH;
            red[(wave * TP + tl) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __syncthreads();
        // 2 halves x TP tiles x 64 lanes float4 sums over the 4 row groups: 512 TP / 4 ... one per thread when TP == 4
        for (int f = tid; f < 2 * TP * 64; f += TPB) {
            const int ln = f & 63, tl = (f >> 6) % TP, h = (f >> 6) / TP;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = red[((2 * q + h) * TP + tl) * 64 + ln];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const int t = t0 + tl, ua = t / NBH, ub = h * NBH + t % NBH;
            const int lc = ln & 15, lg = ln >> 4;
            const int cbq = col_of<WB>(ub, lc);
            const float sv[4] = { s.x, s.y, s.z, s.w };
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (!RAG || cbq < KR) atomicAdd(DW + (size_t)col_of<WA>(ua, 4 * lg + r) * (RAG ? KR : WB) + cbq, sv[r]);
        }
    }
    if (DB) {
        // column sums of dz: over g inside the wave, then over the waves that hold them
        constexpr int NZ = LAYOUT == 0 ? NA : NBH;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NZ; ++u) {
            float v = zsum[u];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (g == 0) {
                const int col = LAYOUT == 0 ? col_of<O>(u, c) : col_of<O>(hb * NBH + u, c);
                lds[wave * O + col] = v;                   // layout 0: only the hb == 0 waves summed; layout 1: each wave its half of the columns
            }
        }
        __syncthreads();
        if (tid < O) {
            float s = 0.0f;
            if constexpr (LAYOUT == 0) {
#pragma unroll
                for (int w = 0; w < NW; w += 2) s += lds[w * O + tid];
            } else {
                // column tid belongs to half h = tile / NBH of the permuted tiling
                const int u = O == 32 ? (tid & 1) : 4 * (tid >> 6) + (tid & 3);
                const int h = u / NBH;
#pragma unroll
                for (int q = 0; q < 4; ++q) s += lds[(2 * q + h) * O + tid];
            }
            atomicAdd(DB + tid, s);
        }
    }
}

template <int KT, int OT, int ACT, int LAYOUT, bool RAG = false, int WO = 0>
static int launch(long R, int KR, const float *x, const float *w, const float *dy, const float *dy2, const float *y, float *dx, float *dw, float *db,
                  hipStream_t st, const float *obs = nullptr, int KO = 0, float *dw1 = nullptr, float *db1 = nullptr) {
    constexpr int K = 16 * KT, O = 16 * OT;
    const size_t lds = ((size_t)2 * ROWS * (K + O + WO) + (ACT ? (size_t)ROWS * O : 0) + (dy2 ? (size_t)ROWS * O : 0)) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    static unsigned long long attr = 0;
    if (cm::dev_first(attr)) {
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&bwd_kernel<KT, OT, ACT, LAYOUT, RAG, WO>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int n_cu = cm::cu_count();
    const long chunks = (R + ROWS - 1) / ROWS;
    // One workgroup per CU for the large batches.  A small batch is spread over FEWER workgroups: every workgroup ends with one
    // float atomic per weight-gradient element (4 - 8 k of them on the same addresses), ~0.1 us of serialised L2 atomics per
    // workgroup against ~3 us per 64-row chunk streamed - the sum is least at ~sqrt(29 chunks) workgroups (measured on MI355X,
    // tools/lin_bwd_sweep.py: 10 000 rows of 64 -> 64: 17 us with 64 workgroups, 32 us with one per chunk)
    int blocks = (int)std::min<long>(std::min<long>(chunks, n_cu), std::max<long>(1, std::lround(std::sqrt(29.0 * (double)chunks))));
    static const int force = [] { const char *e = getenv("COMMARL_LIN2_BLOCKS"); return e ? atoi(e) : 0; }();
    if (force > 0) blocks = (int)std::min<long>(chunks, force);
    hipLaunchKernelGGL((bwd_kernel<KT, OT, ACT, LAYOUT, RAG, WO>), dim3(blocks), dim3(TPB), std::max<size_t>(lds, 33 * 1024), st, R, KR, x, w, dy, dy2, y, dx, dw,
                       db, obs, KO, dw1, db1);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

}  // namespace lin2

// Returns 1 when this shape / alignment is not covered (the caller runs lin::bwd_kernel), else the launch status.
int linear_bwd_stream(long R, int K, int O, const float *x, const float *w, int layout, const float *dy, const float *dy2,
                      const float *y, float *dx, float *dw, float *db, void *stream) {
    static const bool off = [] { const char *e = getenv("COMMARL_LIN_BWD"); return e && e[0] == 'o'; }();   // "old"
    if (off) return 1;
    const auto ok_w = [](int v) { return v == 32 || v == 64 || v == 128; };
    const hipStream_t st = (hipStream_t)stream;
    if (!ok_w(K) && ok_w(O) && K <= 128 && !dx && layout == 0 && !(((uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)y) & 15) && !((uintptr_t)x & 3)) {
        // first layer (observation -> hidden): ragged input rows, no input gradient
#define CM_RG(KT_, OT_) (y ? lin2::launch<KT_, OT_, 1, 0, true>(R, K, x, w, dy, dy2, y, dx, dw, db, st) : lin2::launch<KT_, OT_, 0, 0, true>(R, K, x, w, dy, dy2, y, dx, dw, db, st))
        const int kt = K <= 32 ? 2 : (K <= 64 ? 4 : 8);
        switch (kt * 1000 + O) {
        case 2032: return CM_RG(2, 2);
        case 2064: return CM_RG(2, 4);
        case 2128: return CM_RG(2, 8);
        case 4032: return CM_RG(4, 2);
        case 4064: return CM_RG(4, 4);
        case 4128: return CM_RG(4, 8);
        case 8032: return CM_RG(8, 2);
        case 8064: return CM_RG(8, 4);
        case 8128: return CM_RG(8, 8);
        default: return 1;
        }
#undef CM_RG
    }
    if (!ok_w(K) || !ok_w(O) || (K == 128 && O == 128)) return 1;
    const uintptr_t al = (uintptr_t)x | (uintptr_t)dy | (uintptr_t)dy2 | (uintptr_t)y | (uintptr_t)dx;
    if (al & 15) return 1;
    if (layout == 1 && !(K == 64 && O == 64)) return 1;   // the [in][out] weights are the 64 x 64 graph-convolution ones
#define CM_B2(KT_, OT_, L_) (y ? lin2::launch<KT_, OT_, 1, L_>(R, K, x, w, dy, dy2, y, dx, dw, db, st) : lin2::launch<KT_, OT_, 0, L_>(R, K, x, w, dy, dy2, y, dx, dw, db, st))
    if (layout == 1) return CM_B2(4, 4, 1);
    switch (K * 1000 + O) {
    case 32032: return CM_B2(2, 2, 0);
    case 32064: return CM_B2(2, 4, 0);
    case 32128: return CM_B2(2, 8, 0);
    case 64032: return CM_B2(4, 2, 0);
    case 64064: return CM_B2(4, 4, 0);
    case 64128: return CM_B2(4, 8, 0);
    case 128032: return CM_B2(8, 2, 0);
    case 128064: return CM_B2(8, 4, 0);
    default: return 1;
    }
#undef CM_B2
}

// Encoder backward in one pass (obs [R,d] -> a1 = tanh(.) [R,128] -> e = tanh(.) [R,64]): layer 2 as linear_bwd_stream
// with dz = (dy + dy2) * (1 - e^2), its input gradient chained in LDS into layer 1's weight / bias gradient.
// Returns 1 when the shape is not covered (d > 64, unaligned tensors): the caller runs the two layers one by one.
int encoder_bwd_chain(long R, int d, const float *obs, const float *a1, const float *e, const float *w2, const float *dy, const float *dy2,
                      float *dw2, float *db2, float *dw1, float *db1, void *stream) {
    static const bool off = [] { const char *v = getenv("COMMARL_ENC_CHAIN"); return v && v[0] == '0'; }();
    if (off || d < 1 || d > 64) return 1;
    if (((uintptr_t)a1 | (uintptr_t)e | (uintptr_t)dy | (uintptr_t)dy2) & 15) return 1;
    if ((uintptr_t)obs & 3) return 1;
    const hipStream_t st = (hipStream_t)stream;
    if (d <= 32) return lin2::launch<8, 4, 1, 0, false, 32>(R, 128, a1, w2, dy, dy2, e, nullptr, dw2, db2, st, obs, d, dw1, db1);
    return lin2::launch<8, 4, 1, 0, false, 64>(R, 128, a1, w2, dy, dy2, e, nullptr, dw2, db2, st, obs, d, dw1, db1);
}

}  // namespace cm
