// Streamed-weights halo kernel: 3x3 stride-1 convolutions with MANY channels
// (conv2_2 .. conv7_1 and the ConvGRU of upstream Backbone.py / V2VNet.py; code absent from
// /root/reference, see include/v2x_amd.h).
//
// The gather kernel (conv_igemm.hip) moves 32 KiB through L2->LDS per 2.1 MFLOP (64 FLOP/B) and tops
// out near 8.7 TB/s of LDS-DMA traffic (~600-800 TFLOP/s).  Here a workgroup owns a 256-pixel spatial
// tile x BCO output channels and walks K as (32-channel chunk) x (9 taps):
//   * the (TH+2)x(TW+2) input patch of a channel chunk is DMA'd into LDS ONCE and serves all 9 taps
//     (for the x2-upsampled decoder source the patch is the half-resolution one: upsample + concat are
//     address arithmetic on patch coordinates);
//   * the weights of one (chunk, tap) are an 8 KiB slice [4 k-slots][BCO][8] streamed through a 4-slot
//     LDS ring, prefetch distance 3, with COUNTED s_waitcnt vmcnt(N) and raw s_barrier (the compiler's
//     __syncthreads would drain the DMA queue every step -- cdna_hip_programming.md, "Pipelining across
//     barriers");  every step a wave issues exactly NW weight DMAs + 1 patch DMA, so the counts are static;
//   * per step a wave does (BCO/16) x 4 MFMAs 16x16x32 from BCO/16 + 4 ds_read_b128 -> ~200 FLOP per
//     L2 byte, 3x the gather kernel.
// LDS: 4 x 8 KiB ring + 2 x 24 KiB patch buffers = 80 KiB -> two workgroups per CU (160 KiB).
// LDS layouts as conv_halo.hip (conflict-free: patch slot ^ ((x>>1)&3), weights k-slot-major).
#include "conv_stream.h"
#include <cstdlib>
#include <type_traits>



// The timing experiments on these kernels (phase-removal builds, per-phase time stamps: tools/stream8_phase_probe.sh, tools/stream8g_timeline.sh) are built
// from an instrumented COPY of this file, tools/probes/conv_stream_probe.hip (`make PROBE=conv_stream`); this file carries none of that scaffolding.
// Patch swizzle: pixel pc's channel slot s (16 bytes) lives at physical slot s ^ PSWZ(pc) of its 64 bytes.  A pixel-fragment read is
// lane (fj, fq) -> pixel pc0 + fj (full resolution) or ((c + fj) >> 1) + 1 (half resolution), slot fq.  tools/lds_conflict_probe.hip times
// candidate swizzles with eight reads in flight: (pc >> 2) & 3 serves every alignment of both forms at 17.5 ns per read, the round-1 choice
// (pc >> 1) & 3 at 22.5 ns for EVERY full-resolution read and for two of three half-resolution alignments -- although SQ_LDS_BANK_CONFLICT
// reads 0 for the full-resolution case (the counter does not see whatever pairing rule ds_read_b128 applies).  INSIDE the kernels the
// faster reads lose: paired in-process A/B (tools/ab_inproc.sh, s.e. 0.1-0.2 %): every streamed layer is 1.0-3.0 % SLOWER with
// (pc >> 2) & 3 (conv3_2 289 -> 298 us, ConvGRU 1 570 -> 1 611 us).  Default = the round-1 swizzle (1); 2 builds the other one.
#ifndef V2X_STREAM_PSWZ_BUILD
#define V2X_STREAM_PSWZ_BUILD 1
#endif
#define PSWZ(pc) (((pc) >> V2X_STREAM_PSWZ_BUILD) & 3)
// Variants of stream8g that were built, measured and REMOVED from this file in round 4 (HISTORY.md "Round 3 -- measured and rejected" holds the
// numbers, the code is in the git history up to commit 7e0187e): the 32x32x16-MFMA form (M32: bit-identical, 9-20 % slower), weight fragments
// read two blocks ahead (PF = 2: +2.8-3.7 %), s_setprio in the load / MFMA phase (LPRIO, PRIO: no effect), the prefetch point inside a block
// (H1: 0.0 %), one channel tile per XCD (XCDCO: +0.1-0.7 %), fragment reads before the DMA issue (LORDER: +0.2-0.5 %).  What moves these kernels
// is the NUMBER of instructions per MFMA (taps per barrier, fragment reads per wave tile), not where their latencies fall.
// Round 4: the one-wave-per-SIMD 32x32x16 form ("w1": 4 waves x 512 registers, each wave issuing its own reads and DMAs between its MFMAs) --
// bit-identical, 23-35 % slower, and with all companion work compiled out only as fast as stream8g with all of its: the chip is power-limited
// on these operands (profiles/r04_w1_rejected.txt; the kernel is conv_stream_w1.hip at commit a173db3).
#ifndef V2X_STREAM_LSS_BUILD
#define V2X_STREAM_LSS_BUILD 1
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- epilogue shared by the 4-wave and the 8-wave kernels -------------------------------------------
// Chained epilogue with its second-GEMM operands staged in LDS (stage_chain): [w2 in fragment order (i2, ks, fq, fj) x 16 B |
// scale2 | shift2].  vmcnt is in-order: a global load issued between the stores of two output-channel tiles can only be
// waited for together with those stores, and the chain needs NKS weight fragments + two vectors per channel tile.
template <int BCO>
constexpr int chain_lds_bytes() { return BCO * BCO * 2 + 2 * BCO * 4; }

template <int BCO, int NT>
__device__ __forceinline__ void stage_chain(const StreamArgs &a, char *cl) {
    constexpr int NKS = BCO / 32;
    for (int p = threadIdx.x; p < BCO * BCO / 8; p += NT) {
        const int row = p / (BCO / 8), c8 = p - row * (BCO / 8);
        const int i2 = row >> 4, fj = row & 15, ks = c8 >> 2, fq = c8 & 3;
        *reinterpret_cast<uint4 *>(cl + (((i2 * NKS + ks) * 4 + fq) * 16 + fj) * 16) =
            *reinterpret_cast<const uint4 *>(a.w2 + (size_t)row * BCO + c8 * 8);
    }
    float *ss = reinterpret_cast<float *>(cl + BCO * BCO * 2);
    for (int i = threadIdx.x; i < BCO; i += NT) {
        ss[i] = a.scale2[i];
        ss[BCO + i] = a.shift2[i];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // before the counted-DMA regime starts; read after many barriers
}

// lss != nullptr (plain and GRU epilogues): scale / shift (or the GRU's bias quadruples) of the rows this wave owns, staged in LDS by the
// kernel -- lss[c] = scale, lss[lss_stride + c] = shift for local row c (GRU: float4 per local hidden channel).  vmcnt is in order: read
// from global memory these few bytes can only be waited for together with every DMA the load phases have in flight, and in the one-workgroup-
// per-CU ping-pong kernels nobody fills that gap (the other group is parked at the barrier).
// PKRELU: ReLU on the packed bf16 pairs (one instruction per two values) instead of fmaxf on the fp32 values (two per value); identical
// results.  Paired A/B, all bit-identical: streamed layers -0.3 ... -0.7 %, 64 -> 64 streamed -2.3 %, chained 128-row layer -1.9 %; only
// conv3x3_wide3_kernel measured +1.6 % with it and passes false.
// X4 (plain epilogue, with x4 = true at run time): 16-byte output stores.  A lane's accumulators of channel tile i are 4 consecutive channels
// (8 bytes) of one pixel, so the plain path issues one dwordx2 store per (tile, fragment) -- 256 per workgroup and 16 x 32 tile, and a vector
// store costs the memory pipeline about the same whatever its width: tools/tile_overhead.py measures ~12 us per tile beyond the K loop on the
// 128-row layers (37 % of conv6_2's time).  v_permlane16_swap_b32 (gfx950) swaps the odd 16-lane rows of one register with the even rows of
// another: applied to the packed results of tiles i and i + 1 it leaves the lanes of even k-slot quarter fq with 8 consecutive channels of
// tile i (theirs + their neighbour's) and the odd ones with 8 consecutive channels of tile i + 1 -- ONE dwordx4 store per tile pair and
// fragment, same bytes, same values.  Needs 16-byte aligned rows (out_cstride, out_coff multiples of 8 channels) and Cout % 32 == 0 (host: a.x4).
template <int BCO, int TW, int EPI, int NF = 4, bool CL = false, bool PKRELU = true, bool X4 = false>
__device__ __forceinline__ void stream_epilogue(const StreamArgs &a, f32x4_t (&acc)[BCO / 16][NF], int co_tile, int n, int y0,
                                                int x0, const int (&frow)[NF], int fj, int fq, const char *cl = nullptr,
                                                lds_cf_t *lss = nullptr, int lss_stride = 0, bool x4 = false) {
    constexpr int TCO = BCO / 16;
    if constexpr (EPI == SEPI_GRU) {
        // rows are (r,z,n) triples of 16 hidden channels: tiles 3g, 3g+1, 3g+2  (packing.pack_gru_stream)
        auto gate = [&](int g, int f, const float4 (&bias)[4], uint2 &o) __attribute__((always_inline)) {
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                h[r] = v2x_gru_h0(acc[3 * g][f][r], acc[3 * g + 1][f][r], acc[3 * g + 2][f][r], bias[r]);
            }
            o.x = pack_bf16x2(h[0], h[1]);
            o.y = pack_bf16x2(h[2], h[3]);
        };
        if constexpr (X4 && (TCO / 3) % 2 == 0) {
            if (x4) {   // 16-byte stores: the hidden-channel groups g, g + 1 exchanged between the k-slot quarters (see the plain path)
#pragma unroll
                for (int g = 0; g < TCO / 3; g += 2) {
                    const int hb = (co_tile * (TCO / 3) + g) * 16 + fq * 4;
                    float4 bias[2][4];
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            bias[h2][r] = lss ? lds_ld4(lss + ((g + h2) * 16 + fq * 4 + r) * 4) : reinterpret_cast<const float4 *>(a.scale)[hb + h2 * 16 + r];
                    const int hc = hb + ((fq & 1) ? 12 : 0);
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        uint2 o0, o1;
                        gate(g, f, bias[0], o0);
                        gate(g + 1, f, bias[1], o1);
                        const auto rx = __builtin_amdgcn_permlane16_swap(o0.x, o1.x, false, false);
                        const auto ry = __builtin_amdgcn_permlane16_swap(o0.y, o1.y, false, false);
                        const size_t pix = (size_t)(n * a.H + y0 + frow[f]) * a.W + x0 + ((TW == 32) ? (f & 1) * 16 + fj : fj);
                        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(a.out) + pix * a.out_cstride + a.out_coff + hc) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int g = 0; g < TCO / 3; ++g) {
            const int hc = (co_tile * (TCO / 3) + g) * 16 + fq * 4;
            if (hc >= a.Cout) continue;
            float4 bias[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                bias[r] = lss ? lds_ld4(lss + (g * 16 + fq * 4 + r) * 4) : reinterpret_cast<const float4 *>(a.scale)[hc + r];
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float h[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    h[r] = v2x_gru_h0(acc[3 * g][f][r], acc[3 * g + 1][f][r], acc[3 * g + 2][f][r], bias[r]);
                }
                uint2 o;
                o.x = pack_bf16x2(h[0], h[1]);
                o.y = pack_bf16x2(h[2], h[3]);
                const size_t pix = (size_t)(n * a.H + y0 + frow[f]) * a.W + x0 + ((TW == 32) ? (f & 1) * 16 + fj : fj);
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(a.out) + pix * a.out_cstride + a.out_coff + hc) = o;
            }
        }
    } else if constexpr (EPI == SEPI_CHAIN) {
        // conv (BCO = ALL channels of the layer, rows in kappa order) -> BN/ReLU -> bf16 -> 1x1 conv BCO -> BCO -> BN/ReLU.
        // Same register-layout trick as conv_halo.hip: tiles (2s, 2s+1) of a lane ARE its B fragment of k-step s.
        // BCO = 64: conv1_2 -> conv3d_1;  BCO = 128: conv2_2 -> conv3d_2.
        static_assert(EPI != SEPI_CHAIN || BCO == 64 || BCO == 128, "chain epilogue needs all channels in one workgroup");
        constexpr int NKS = BCO / 32;   // k-steps of the second GEMM
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // leave the counted-DMA regime before ordinary loads
        // hidden activations of all 4 pixel fragments as B fragments (bf16): NKS x 4 registers each.  k-step outermost so
        // that only two tiles' scale/shift are live and the accumulator tiles die as they are consumed (register pressure).
        bf16x8_t hb[NF][NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            float4 sc[2], sf[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int i = 2 * ks + hf;
                const int kappa = 32 * (i >> 1) + 8 * fq + 4 * (i & 1);
                sc[hf] = *reinterpret_cast<const float4 *>(a.scale + kappa);
                sf[hf] = *reinterpret_cast<const float4 *>(a.shift + kappa);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float h[8];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int i = 2 * ks + hf;
                    h[hf * 4 + 0] = acc[i][f][0] * sc[hf].x + sf[hf].x;
                    h[hf * 4 + 1] = acc[i][f][1] * sc[hf].y + sf[hf].y;
                    h[hf * 4 + 2] = acc[i][f][2] * sc[hf].z + sf[hf].z;
                    h[hf * 4 + 3] = acc[i][f][3] * sc[hf].w + sf[hf].w;
                }
                uint4 p;
                p.x = pack_bf16x2(h[0], h[1]);
                p.y = pack_bf16x2(h[2], h[3]);
                p.z = pack_bf16x2(h[4], h[5]);
                p.w = pack_bf16x2(h[6], h[7]);
                if (a.relu) {   // ReLU on the packed pairs (v2x_relu_bf16x2: one instruction per two values)
                    p.x = v2x_relu_bf16x2(p.x);
                    p.y = v2x_relu_bf16x2(p.y);
                    p.z = v2x_relu_bf16x2(p.z);
                    p.w = v2x_relu_bf16x2(p.w);
                }
                hb[f][ks] = __builtin_bit_cast(bf16x8_t, p);
            }
        }
        size_t pix[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
            pix[f] = (size_t)(n * a.H + y0 + frow[f]) * a.W + x0 + ((TW == 32) ? (f & 1) * 16 + fj : fj);
        // second GEMM: one output-channel tile at a time; its NKS weight fragments (L1/L2-resident, BCO*BCO*2 bytes in
        // all) are loaded once and serve the 4 pixel fragments
        auto chain_tile = [&](int i2, bf16x8_t (&w2f)[NKS], float4 &s2, float4 &t2) __attribute__((always_inline)) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                w2f[ks] = CL ? *reinterpret_cast<const bf16x8_t *>(cl + (((i2 * NKS + ks) * 4 + fq) * 16 + fj) * 16)
                             : *reinterpret_cast<const bf16x8_t *>(a.w2 + (size_t)(i2 * 16 + fj) * BCO + ks * 32 + fq * 8);
            const int co = i2 * 16 + fq * 4;
            s2 = *reinterpret_cast<const float4 *>(CL ? reinterpret_cast<const float *>(cl + BCO * BCO * 2) + co : a.scale2 + co);
            t2 = *reinterpret_cast<const float4 *>(CL ? reinterpret_cast<const float *>(cl + BCO * BCO * 2) + BCO + co : a.shift2 + co);
        };
        if constexpr (X4) {
            if (x4) {   // 16-byte stores: output tiles i2, i2 + 1 exchanged between the k-slot quarters (see the plain path)
                const uint32_t floor2 = a.relu2 ? 0u : 0x80008000u;
#pragma unroll
                for (int i2 = 0; i2 < TCO; i2 += 2) {
                    bf16x8_t w2f[2][NKS];
                    float4 s2[2], t2[2];
                    chain_tile(i2, w2f[0], s2[0], t2[0]);
                    chain_tile(i2 + 1, w2f[1], s2[1], t2[1]);
                    const int co = i2 * 16 + fq * 4 + ((fq & 1) ? 12 : 0);
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        uint32_t ox[2], oy[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int ks = 0; ks < NKS; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[h][ks], hb[f][ks], d, 0, 0, 0);
                            ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(d[0] * s2[h].x + t2[h].x, d[1] * s2[h].y + t2[h].y), floor2);
                            oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(d[2] * s2[h].z + t2[h].z, d[3] * s2[h].w + t2[h].w), floor2);
                        }
                        const auto rx = __builtin_amdgcn_permlane16_swap(ox[0], ox[1], false, false);
                        const auto ry = __builtin_amdgcn_permlane16_swap(oy[0], oy[1], false, false);
                        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(a.out) + pix[f] * a.out_cstride + a.out_coff + co) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int i2 = 0; i2 < TCO; ++i2) {
            bf16x8_t w2f[NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                w2f[ks] = CL ? *reinterpret_cast<const bf16x8_t *>(cl + (((i2 * NKS + ks) * 4 + fq) * 16 + fj) * 16)
                             : *reinterpret_cast<const bf16x8_t *>(a.w2 + (size_t)(i2 * 16 + fj) * BCO + ks * 32 + fq * 8);
            const int co = i2 * 16 + fq * 4;
            const float4 s2 = *reinterpret_cast<const float4 *>(CL ? reinterpret_cast<const float *>(cl + BCO * BCO * 2) + co : a.scale2 + co);
            const float4 t2 = *reinterpret_cast<const float4 *>(CL ? reinterpret_cast<const float *>(cl + BCO * BCO * 2) + BCO + co : a.shift2 + co);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                f32x4_t d = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks], hb[f][ks], d, 0, 0, 0);
                const float v0 = d[0] * s2.x + t2.x, v1 = d[1] * s2.y + t2.y, v2 = d[2] * s2.z + t2.z, v3 = d[3] * s2.w + t2.w;
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                if (a.relu2) {
                    o.x = v2x_relu_bf16x2(o.x);
                    o.y = v2x_relu_bf16x2(o.y);
                }
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(a.out) + pix[f] * a.out_cstride + a.out_coff + co) = o;
            }
        }
    } else {
        if constexpr (X4 && TCO % 2 == 0) {
            if (x4) {
                const uint32_t floor_bits = a.relu ? 0u : 0x80008000u;      // v2x_relu_bf16x2_floor: ReLU or identity, one instruction per pair either way
#pragma unroll
                for (int i = 0; i < TCO; i += 2) {
                    const int cb = co_tile * BCO + i * 16 + fq * 4;         // this lane's channels in tile i (tile i + 1: + 16)
                    float4 sc[2], sf[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        sc[h] = lss ? lds_ld4(lss + (i + h) * 16 + fq * 4) : *reinterpret_cast<const float4 *>(a.scale + cb + h * 16);
                        sf[h] = lss ? lds_ld4(lss + lss_stride + (i + h) * 16 + fq * 4) : *reinterpret_cast<const float4 *>(a.shift + cb + h * 16);
                    }
                    const int co = cb + ((fq & 1) ? 12 : 0);                // even quarters: 8 channels of tile i from cb; odd: of tile i + 1 from cb + 16 - 4
#pragma unroll
                    for (int f = 0; f < NF; ++f) {
                        uint32_t ox[2], oy[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            ox[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][0] * sc[h].x + sf[h].x, acc[i + h][f][1] * sc[h].y + sf[h].y), floor_bits);
                            oy[h] = v2x_relu_bf16x2_floor(pack_bf16x2(acc[i + h][f][2] * sc[h].z + sf[h].z, acc[i + h][f][3] * sc[h].w + sf[h].w), floor_bits);
                        }
                        const auto rx = __builtin_amdgcn_permlane16_swap(ox[0], ox[1], false, false);
                        const auto ry = __builtin_amdgcn_permlane16_swap(oy[0], oy[1], false, false);
                        const size_t pix = (size_t)(n * a.H + y0 + frow[f]) * a.W + x0 + ((TW == 32) ? (f & 1) * 16 + fj : fj);
                        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(a.out) + pix * a.out_cstride + a.out_coff + co) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                    }
                }
                return;
            }
        }
#pragma unroll
        for (int i = 0; i < TCO; ++i) {
            const int co = co_tile * BCO + i * 16 + fq * 4;
            if (co >= a.Cout) continue;
            const float4 sc = lss ? lds_ld4(lss + i * 16 + fq * 4) : *reinterpret_cast<const float4 *>(a.scale + co);
            const float4 sf = lss ? lds_ld4(lss + lss_stride + i * 16 + fq * 4) : *reinterpret_cast<const float4 *>(a.shift + co);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float v0 = acc[i][f][0] * sc.x + sf.x, v1 = acc[i][f][1] * sc.y + sf.y;
                float v2 = acc[i][f][2] * sc.z + sf.z, v3 = acc[i][f][3] * sc.w + sf.w;
                if (!PKRELU && a.relu) {
                    v0 = fmaxf(v0, 0.f);
                    v1 = fmaxf(v1, 0.f);
                    v2 = fmaxf(v2, 0.f);
                    v3 = fmaxf(v3, 0.f);
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                if (PKRELU && a.relu) {   // on the packed bf16 pairs: 2 instructions per 4 values (fmaxf on the fp32 values: 8 -- 256 of a wave's ~600 epilogue instructions)
                    o.x = v2x_relu_bf16x2(o.x);
                    o.y = v2x_relu_bf16x2(o.y);
                }
                const size_t pix = (size_t)(n * a.H + y0 + frow[f]) * a.W + x0 + ((TW == 32) ? (f & 1) * 16 + fj : fj);
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(a.out) + pix * a.out_cstride + a.out_coff + co) = o;
            }
        }
    }
}

template <int BCO, int TH, int TW, int EPI, bool SPLITK = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_stream_kernel(const StreamArgs a) {
    constexpr int PW = TW + 2, PH = TH + 2, PW0 = TW / 2 + 2, PH0 = TH / 2 + 2;
    constexpr int TCO = BCO / 16;
    constexpr int W_PIECES = BCO / 16;        // 1 KiB pieces per weight slice (BCO rows x 64 B)
    constexpr int SLICE_BYTES = BCO * 64;
    static_assert(PH * PW * 4 <= (PATCH_PIECES - 1) * 64, "patch must fit its buffer and leave the last piece as padding");
    static_assert(TH * TW == 256 && (TW == 32 || TW == 16), "256-pixel tiles: 8x32 or 16x16");
    static_assert(W_PIECES >= 4 && W_PIECES <= 8, "1 or 2 weight DMAs per wave per step");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_ring = smem;                              // RING x SLICE_BYTES
    char *s_patch = smem + RING * SLICE_BYTES;        // 2 x PATCH_BYTES
    char *s_dummy = s_patch + (PATCH_PIECES - 1) * 1024;  // count-keeping dummy DMAs land in patch padding (never read)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;
    const int NW = (wave + 4 < W_PIECES) ? 2 : 1;     // weight DMAs this wave issues per step (wave-uniform)

    // XCD-aware order (see conv_igemm.hip)
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int co_tile = bid % a.n_co_tiles;
    const int px_tile = bid / a.n_co_tiles;
    const int txy = a.tiles_x * a.tiles_y;
    const int n = px_tile / txy;
    const int trem = px_tile - n * txy;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    const int nc0 = a.C0 >> 5, nchunks = (a.C0 + a.C1) >> 5;
    // the chunk range of this workgroup: all of them, or (SPLITK) the blockIdx.y-th of a.ksplit contiguous ranges (the host makes sure
    // that none is empty); steps and ring slots are counted from the range's first step
    int kc_lo = 0, kc_hi = nchunks;
    if constexpr (SPLITK) {
        const int per = (nchunks + a.ksplit - 1) / a.ksplit;
        kc_lo = blockIdx.y * per;
        kc_hi = kc_lo + per < nchunks ? kc_lo + per : nchunks;
    }
    const int S = (kc_hi - kc_lo) * 9;
    const uint16_t *wbase = a.w + ((size_t)co_tile * nchunks + kc_lo) * 9 * (BCO * 32);
    // zero page for out-of-image patch pixels and count-keeping dummy DMAs: the packer appends 64 B of zeros
    const void *zero_page = a.w + (size_t)a.n_co_tiles * nchunks * 9 * (BCO * 32);

    // ---- per-lane tables, computed once (the step body then only adds and selects) ---------------------
    // (a) DMA source descriptors of this lane's 6 patch pieces (piece = wave + 4t), for a full-resolution and a
    //     half-resolution source:  (source pixel index << 5) | (swizzled 16-B slot * 8 elements),  -1 = zero page
    int pd_full[6], pd_half[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int L = (wave + 4 * t) * 64 + lane;
        const int pix = L >> 2, phys = L & 3;
        {
            const int pr = pix / PW, pc = pix - pr * PW;
            const int y = y0 - 1 + pr, x = x0 - 1 + pc;
            const bool ok = pix < PH * PW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            pd_full[t] = ok ? ((((n * a.H + y) * a.W + x) << 5) | ((phys ^ PSWZ(pc)) << 3)) : -1;
        }
        {
            const int Hs = a.H >> 1, Ws = a.W >> 1;
            const int pr = pix / PW0, pc = pix - pr * PW0;
            const int y = (y0 >> 1) - 1 + pr, x = (x0 >> 1) - 1 + pc;
            const bool ok = pix < PH0 * PW0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
            pd_half[t] = ok ? ((((n * Hs + y) * Ws + x) << 5) | ((phys ^ PSWZ(pc)) << 3)) : -1;
        }
    }
    // (b) byte offset inside a patch row of this lane's B fragment f for tap column kx (pixel slot + swizzled k-slot)
    int frow[4], ct_full[4][3], ct_half[4][3];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const int col = (TW == 32) ? (f & 1) * 16 + fj : fj;
        frow[f] = (TW == 32) ? 2 * wave + (f >> 1) : 4 * wave + f;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int pcf = col + kx;                    // full res: column col + kx
            const int pch = ((col + kx - 1) >> 1) + 1;   // half res: floor((col + kx - 1) / 2) + 1
            ct_full[f][kx] = ((pcf << 2) + (fq ^ PSWZ(pcf))) * 16;
            ct_half[f][kx] = ((pch << 2) + (fq ^ PSWZ(pch))) * 16;
        }
    }

    // ---- DMA issue helpers -------------------------------------------------------------------------
    const uint16_t *wsrc = wbase + lane * 8 + wave * 512;  // this lane's element of slice 0, piece `wave`
    auto issue_weights = [&](int s) {  // slice of step s -> ring slot s % RING ; NW instructions
        char *dst = s_ring + (s & (RING - 1)) * SLICE_BYTES;
        const uint16_t *src = wsrc + (size_t)s * (BCO * 32);
        glds16s(src, dst + wave * 1024);
        if (NW == 2) glds16s(src + 4 * 512, dst + (wave + 4) * 1024);
    };
    auto issue_patch_piece = [&](int kc, int t, int buf) {  // piece wave+4t of chunk kc's patch
        const bool first = kc < nc0;
        const bool hf = first && a.up0;
        // 6-way select by the wave-uniform t (a runtime-indexed register array would go to scratch)
        int d = hf ? pd_half[0] : pd_full[0];
#pragma unroll
        for (int u = 1; u < 6; ++u) d = (t == u) ? (hf ? pd_half[u] : pd_full[u]) : d;
        const uint16_t *src = first ? a.in0 : a.in1;
        const unsigned cs = first ? (unsigned)a.C0 : (unsigned)a.C1;
        const unsigned off = (unsigned)(d >> 5) * cs + (unsigned)((first ? kc : kc - nc0) * 32 + (d & 31));
        glds16s(d >= 0 ? (const void *)(src + off) : zero_page, s_patch + buf * PATCH_BYTES + (wave + 4 * t) * 1024);
    };
    auto issue_dummy = [&]() { glds16s(zero_page, s_dummy); };

    // ---- prologue: whole patch of chunk 0 (6 pieces per wave), weight slices of steps 0..2 --------
#pragma unroll
    for (int t = 0; t < 6; ++t) issue_patch_piece(kc_lo, t, 0);
    issue_weights(0);  // S = 9 * chunks >= 9, so steps 1 and 2 always exist
    issue_weights(1);
    issue_weights(2);

    f32x4_t acc[TCO][4];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    int s = 0;
    for (int kc = kc_lo; kc < kc_hi; ++kc) {
        const char *pb = s_patch + ((kc - kc_lo) & 1) * PATCH_BYTES;
        const bool half = (kc < nc0) && a.up0;
        const int sh = half ? 1 : 0, row_bytes = (half ? PW0 : PW) * 64;  // wave-uniform
#pragma unroll 1  // ky stays rolled (full unrolling hoists per-tap offsets into ~70 registers and spills; scratch
                  // traffic would also corrupt the vmcnt bookkeeping); kx is unrolled so the table indices are static
        for (int ky = 0; ky < 3; ++ky) {
            // patch row of each fragment for this ky: full res row+ky, half res floor((row+ky-1)/2)+1
            int rowoff[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) rowoff[f] = (((frow[f] + ky - sh) >> sh) + sh) * row_bytes;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx, ++s) {
                const int tap = ky * 3 + kx;
                // 1. wait for this step's weight slice (and, at tap 0, the chunk's patch): everything older than
                //    the two most recent groups has landed.  group = NW weight DMAs + 1 patch/dummy DMA.
                if (s + 3 >= S) wait_vmcnt<0>();              // tail: nothing (or not everything) is issued below
                else if (s < 2) { if (NW == 2) wait_vmcnt<4>(); else wait_vmcnt<2>(); }
                else { if (NW == 2) wait_vmcnt<6>(); else wait_vmcnt<4>(); }
                // 2. everyone's pieces have landed; everyone is done reading ring slot (s-1) and, at tap 0, the
                //    other patch buffer
                __builtin_amdgcn_s_barrier();
                // 3. keep the pipe full: weight slice of step s+3, one piece of the next chunk's patch
                if (s + 3 < S) {
                    issue_weights(s + 3);
                    if (tap < 6 && kc + 1 < kc_hi) issue_patch_piece(kc + 1, tap, (kc + 1 - kc_lo) & 1);
                    else issue_dummy();
                }
                // 4. MFMAs of (chunk kc, tap)
                const char *ws = s_ring + (s & (RING - 1)) * SLICE_BYTES;
                bf16x8_t fa[TCO], fb[4];
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    fb[f] = *reinterpret_cast<const bf16x8_t *>(pb + rowoff[f] + (half ? ct_half[f][kx] : ct_full[f][kx]));
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    fa[i] = *reinterpret_cast<const bf16x8_t *>(ws + (fq * BCO + i * 16 + fj) * 16);
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[f], acc[i][f], 0, 0, 0);

                // schedule (measured): ALL fragment reads first, one wait, then the 32-MFMA block.  hipcc on its own
                // recycles a single A register set and exposes the LDS latency TCO times per step (-15 %); a finer
                // read/MFMA interleave, s_setprio around the block, and issuing the reads before the DMA-issue block (in-process
                // A/B: 433.9 vs 435.3 us) were not faster: the co-resident wave of the
                // other workgroup is what hides this wave's read phase.
                __builtin_amdgcn_sched_group_barrier(0x100, TCO + 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TCO * 4, 0);
            }
        }
    }

    if constexpr (SPLITK) {
        // raw fp32 partial sums of this chunk range: ws[split][pixel][w_rows], the lane's 4 consecutive rows as one 16-byte store
        const size_t npix = (size_t)a.N * a.H * a.W;
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const size_t pix = (size_t)(n * a.H + y0 + frow[f]) * a.W + x0 + ((TW == 32) ? (f & 1) * 16 + fj : fj);
                *reinterpret_cast<f32x4_t *>(a.ws + ((size_t)blockIdx.y * npix + pix) * a.w_rows + co_tile * BCO + i * 16 + fq * 4) = acc[i][f];
            }
    } else {
        stream_epilogue<BCO, TW, EPI, 4, false, true, true>(a, acc, co_tile, n, y0, x0, frow, fj, fq, nullptr, nullptr, 0, a.x4 != 0);
    }
}

// Second half of a split-K launch: out = epilogue(sum over the splits, in split order -- deterministic) of ws[split][pixel][w_rows].
// Plain layers: a thread owns 4 consecutive channels of a pixel (scale / shift / ReLU, bf16).  ConvGRU: a thread owns 4 hidden channels
// hc..hc+3, whose gate rows sit at tile * 96 + group * 48 + {0, 16, 32} + c of the packed (r, z, n)-triple order (hc = 32 tile + 16 group + c).
template <bool GRU>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ ws, int ksplit, long long npix, int w_rows, int Cout,
                                                            const float *__restrict__ scale, const float *__restrict__ shift, int relu,
                                                            uint16_t *__restrict__ out, int out_cstride, int out_coff) {
    const int quads = Cout / 4;
    const long long total = npix * quads;
    for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
        const long long pix = t / quads;
        const int c = (int)(t - pix * quads) * 4;
        if constexpr (GRU) {
            const int row = (c >> 5) * 96 + ((c >> 4) & 1) * 48 + (c & 15);
            f32x4_t g[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) g[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < ksplit; ++sp)
#pragma unroll
                for (int q = 0; q < 3; ++q) g[q] += *reinterpret_cast<const f32x4_t *>(ws + ((size_t)sp * npix + pix) * w_rows + row + 16 * q);
            float h[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 b = reinterpret_cast<const float4 *>(scale)[c + r];   // (b_ir + b_hr, b_iz + b_hz, b_in, b_hn)
                h[r] = v2x_gru_h0(g[0][r], g[1][r], g[2][r], b);
            }
            uint2 o;
            o.x = pack_bf16x2(h[0], h[1]);
            o.y = pack_bf16x2(h[2], h[3]);
            *reinterpret_cast<uint2 *>(out + pix * out_cstride + out_coff + c) = o;
        } else {
            f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            for (int sp = 0; sp < ksplit; ++sp) v += *reinterpret_cast<const f32x4_t *>(ws + ((size_t)sp * npix + pix) * w_rows + c);
            const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sf = *reinterpret_cast<const float4 *>(shift + c);
            uint2 o;
            o.x = pack_bf16x2(v[0] * sc.x + sf.x, v[1] * sc.y + sf.y);
            o.y = pack_bf16x2(v[2] * sc.z + sf.z, v[3] * sc.w + sf.w);
            if (relu) {
                o.x = v2x_relu_bf16x2(o.x);
                o.y = v2x_relu_bf16x2(o.y);
            }
            *reinterpret_cast<uint2 *>(out + pix * out_cstride + out_coff + c) = o;
        }
    }
}

// ---- 8-wave "ping-pong" form --------------------------------------------------------------------------
// One 512-thread workgroup per CU owns a 16x32-pixel tile: wave group g (waves 4g..4g+3) computes pixel rows
// [8g, 8g+8) with exactly the fragment mapping of the 4-wave kernel, and both groups share ONE weight ring and ONE
// patch, so a weight slice is fetched L2->LDS once per 512 pixels (half the DMA instructions per MFMA: 1 weight piece
// + 1 patch piece per wave and step; an LDS-DMA instruction costs 60-185 issue cycles next to MFMAs,
// MI355X_MICROARCH.md).  The two groups run half a step out of phase:
//
//      interval   2s          2s+1        2s+2
//      group 0    L(s)        M(s)        L(s+1)        L(s) = issue the DMAs of step s+3, read the 12 fragments
//      group 1    M(s-1)      L(s)        M(s)                 of step s, wait for own pieces of slice s+1
//                                                       M(s) = the 32 MFMAs of step s
// with one s_barrier between intervals (group 1 takes one extra barrier up front, group 0 one at the end), so on
// every SIMD the MFMA block of one wave always runs beside the load phase of the other -- the overlap the two
// independent 4-wave workgroups per CU only get by chance.
// Hazards: (R1) a wave waits for its own pieces of slice s+1 (vmcnt(4): two newer 2-DMA groups may be in flight)
// before the barrier closing its L(s); both groups' L(s) close before interval 2s+2, where slice s+1 is first read.
// (R2) slice s+3 overwrites the ring slot of slice s-1, last read in group 1's L(s-1) (interval 2s-1, lgkmcnt(0)
// before its barrier); the first overwrite is issued in interval 2s.  (R3) the next chunk's patch goes to the
// other buffer, last read in L(9kc-1), first written in L(9kc); its last piece is issued at tap 4, five steps
// before it is read.  Every wave issues exactly 2 DMAs per step (dummies keep the count), so the counts are static.

// PERSISTENT over tiles: a workgroup walks tiles bid, bid + gridDim.x, ... of ONE channel tile (the host makes gridDim.x a
// multiple of n_co_tiles), and the K pipeline never drains between tiles: during the last chunk of a tile the idle patch
// buffer receives chunk 0 of the NEXT tile and the weight ring simply wraps to slice 0 (same channel tile = same weights),
// so the next tile's first step finds its operands in LDS.  With one 112-KiB workgroup per CU nothing else could hide the
// per-tile prologue (patch + 3 slices from L2/HBM, ~3-5k cycles) and the workgroup re-dispatch; the short-K layers
// (36-72 steps per tile) lost 20-30 % to it.  The epilogue's output stores are younger than the DMA groups the next two
// steps wait for, so those two steps allow N_ST more operations in flight (vmcnt(4 + N_ST)) instead of stalling on the
// HBM write latency; from the third step on the stores are older than the awaited group and vmcnt(4) is exact again.
template <int BCO, int EPI>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_stream8_kernel(const StreamArgs a) {
    // the chained-1x1 epilogue needs ~250 VGPRs on its own: it stays one tile per workgroup (the tile loop below then
    // runs once and the compiler drops everything that would have to live across the epilogue)
    constexpr bool PERSIST = (EPI != SEPI_CHAIN);
    constexpr int TH = 16, TW = 32;
    constexpr int PW = TW + 2, PH = TH + 2, PW0 = TW / 2 + 2, PH0 = TH / 2 + 2;
    constexpr int TCO = BCO / 16;
    constexpr int W_PIECES = BCO / 16;
    constexpr int SLICE_BYTES = BCO * 64;
    // output-store instructions per wave and tile.  (Only the chained epilogue of this kernel takes the 16-byte form, stream_epilogue X4: the plain
    // 128-row instantiation -- the STREAM_G = 0 comparison form -- spills with both store paths compiled in, and a spill breaks the vmcnt counts.)
    constexpr int N_ST = (EPI == SEPI_GRU) ? (TCO / 3) * 4 : TCO * 4;
    static_assert(PH * PW * 4 <= (PATCH8_PIECES - 1) * 64, "patch must leave the last piece as padding");
    static_assert(W_PIECES <= 8, "at most one weight DMA per wave per step");
    static_assert(4 + N_ST <= 63, "vmcnt is a 6-bit counter");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_ring = smem;
    char *s_patch = smem + RING * SLICE_BYTES;
    char *s_dummy = s_patch + (PATCH8_PIECES - 1) * 1024;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int grp = wave >> 2, wv = wave & 3;
    const int fj = lane & 15, fq = lane >> 4;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    const int co_tile = bid % a.n_co_tiles;        // the same for every tile of this workgroup (nwg % n_co_tiles == 0)
    const int txy = a.tiles_x * a.tiles_y;

    const int nc0 = a.C0 >> 5, nchunks = (a.C0 + a.C1) >> 5;
    const int S = nchunks * 9;
    const uint16_t *wbase = a.w + (size_t)co_tile * nchunks * 9 * (BCO * 32);
    const void *zero_page = a.w + (size_t)a.n_co_tiles * nchunks * 9 * (BCO * 32);
    char *s_chain = smem + RING * SLICE_BYTES + 2 * PATCH8_BYTES;
    if constexpr (EPI == SEPI_CHAIN) stage_chain<BCO, 512>(a, s_chain);

    // tile coordinates
    auto tile_coords = [&](int t, int &n, int &y0, int &x0) {
        const int px_tile = t / a.n_co_tiles;
        n = px_tile / txy;
        const int trem = px_tile - n * txy;
        const int ty = trem / a.tiles_x;
        y0 = ty * TH;
        x0 = (trem - ty * a.tiles_x) * TW;
    };
    // DMA source descriptor of patch piece (wave + 8t) for a tile at (n, y0, x0):
    // (source pixel index << 5) | (swizzled 16-B slot * 8 elements), -1 = zero page
    auto desc = [&](int n, int y0, int x0, int t, bool hf) -> int {
        const int L = (wave + 8 * t) * 64 + lane;
        const int pix = L >> 2, phys = L & 3;
        if (!hf) {
            const int pr = pix / PW, pc = pix - pr * PW;
            const int y = y0 - 1 + pr, x = x0 - 1 + pc;
            const bool ok = pix < PH * PW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            return ok ? ((((n * a.H + y) * a.W + x) << 5) | ((phys ^ PSWZ(pc)) << 3)) : -1;
        }
        const int Hs = a.H >> 1, Ws = a.W >> 1;
        const int pr = pix / PW0, pc = pix - pr * PW0;
        const int y = (y0 >> 1) - 1 + pr, x = (x0 >> 1) - 1 + pc;
        const bool ok = pix < PH0 * PW0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        return ok ? ((((n * Hs + y) * Ws + x) << 5) | ((phys ^ PSWZ(pc)) << 3)) : -1;
    };

    // per-lane fragment tables.  ONE column-offset table, valid for the resolution of the chunk being computed (it is
    // rebuilt at the at most two points of a tile where the source resolution changes): keeping a full- and a
    // half-resolution copy side by side cost 12 VGPRs this kernel does not have (256-VGPR budget, 128 accumulators).
    int frow[4], ct[4][3];
#pragma unroll
    for (int f = 0; f < 4; ++f) frow[f] = 8 * grp + 2 * wv + (f >> 1);
    auto build_ct = [&](bool hf) {
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const int col = (f & 1) * 16 + fj;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int pc = hf ? (((col + kx - 1) >> 1) + 1) : (col + kx);
                ct[f][kx] = ((pc << 2) + (fq ^ PSWZ(pc))) * 16;
            }
        }
    };

    const bool has_w = wave < W_PIECES;                       // wave-uniform
    const uint16_t *wsrc = wbase + lane * 8 + wave * 512;
    auto issue_dummy = [&]() { glds16s(zero_page, s_dummy); };
    auto issue_slice = [&](int slice, int slot, bool real) {  // exactly one DMA
        if (has_w && real) glds16s(wsrc + (size_t)slice * (BCO * 32), s_ring + slot * SLICE_BYTES + wave * 1024);
        else issue_dummy();
    };
    auto issue_piece = [&](int d, int kc, int t, int buf) {   // exactly one DMA; d = descriptor of this lane
        const bool first = kc < nc0;
        const uint16_t *src = first ? a.in0 : a.in1;
        const unsigned cs = first ? (unsigned)a.C0 : (unsigned)a.C1;
        const unsigned off = (unsigned)(d >> 5) * cs + (unsigned)((first ? kc : kc - nc0) * 32 + (d & 31));
        glds16s(d >= 0 ? (const void *)(src + off) : zero_page, s_patch + buf * PATCH8_BYTES + (wave + 8 * t) * 1024);
    };
    const bool chunk0_half = (nc0 > 0) && a.up0;

    int tile = bid;
    int n, y0, x0;
    tile_coords(tile, n, y0, x0);
    // patch descriptors (5 pieces per wave) of the fill that is in progress / comes next; rebuilt when the fill's tile or
    // source resolution changes
    int pd[5];
    bool pd_half = chunk0_half;
#pragma unroll
    for (int t = 0; t < 5; ++t) pd[t] = desc(n, y0, x0, t, pd_half);

    // prologue of the FIRST tile: patch of chunk 0 (5 pieces per wave), then the 2-DMA groups of steps 0, 1, 2
#pragma unroll
    for (int t = 0; t < 5; ++t) issue_piece(pd[t], 0, t, 0);
    issue_slice(0, 0, true); issue_dummy();
    issue_slice(1, 1, true); issue_dummy();
    issue_slice(2, 2, true); issue_dummy();
    wait_vmcnt<4>();                       // patch 0 and slice 0 have landed (groups 1 and 2 may be in flight)
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();   // half-step offset

    int g = 0;        // global step counter: ring slot of step s of any tile = g & (RING - 1)
    int gc = 0;       // global chunk counter: patch buffer = gc & 1
    int relaxed = 0;  // steps left whose wait must tolerate the previous tile's output stores
    for (;;) {
        const int next = tile + nwg;
        const bool has_next = PERSIST && next < n_tiles;
        int nn = 0, ny0 = 0, nx0 = 0;
        if (has_next) tile_coords(next, nn, ny0, nx0);

        f32x4_t acc[TCO][4];
#pragma unroll
        for (int i = 0; i < TCO; ++i)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        int s = 0;
        for (int kc = 0; kc < nchunks; ++kc, ++gc) {
            const char *pb = s_patch + (gc & 1) * PATCH8_BYTES;
            const bool half = (kc < nc0) && a.up0;
            const int sh = half ? 1 : 0, row_bytes = (half ? PW0 : PW) * 64;
            const bool last_chunk = kc + 1 == nchunks;
            // the patch that streams in during this chunk: the tile's next chunk, or chunk 0 of the NEXT tile.  The current
            // tile's descriptors are dead once its last chunk has started, so the tables switch to the next tile here.
            const int kcn = last_chunk ? 0 : kc + 1;
            const bool fill = !last_chunk || has_next;
            const bool hfn = (kcn < nc0) && a.up0;
            if ((last_chunk && has_next) || (fill && hfn != pd_half)) {   // next tile / the source resolution changes
                const int dn = last_chunk ? nn : n, dy = last_chunk ? ny0 : y0, dx = last_chunk ? nx0 : x0;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    pd[t] = desc(dn, dy, dx, t, hfn);
                    __builtin_amdgcn_sched_barrier(0);   // one descriptor at a time: their temporaries sit on top of 176 live registers
                }
                pd_half = hfn;
            }
            if (kc == 0 || kc == nc0) build_ct(half);  // wave-uniform
#pragma unroll 1
            for (int ky = 0; ky < 3; ++ky) {
                int rowoff[4];
#pragma unroll
                for (int f = 0; f < 4; ++f) rowoff[f] = (((frow[f] + ky - sh) >> sh) + sh) * row_bytes;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx, ++s, ++g) {
                    const int tap = ky * 3 + kx;
                    // ---- L(s): DMAs of step s+3 (wrapping into the next tile), fragments of step s
                    {
                        const int s3 = s + 3;
                        issue_slice(s3 < S ? s3 : s3 - S, (g + 3) & (RING - 1), s3 < S || has_next);
                    }
                    if (tap < 5 && fill) {
                        int d = pd[0];
#pragma unroll
                        for (int u = 1; u < 5; ++u) d = (tap == u) ? pd[u] : d;
                        issue_piece(d, kcn, tap, (gc + 1) & 1);
                    } else {
                        issue_dummy();
                    }
                    const char *ws = s_ring + (g & (RING - 1)) * SLICE_BYTES;
                    bf16x8_t fa[TCO], fb[4];
#pragma unroll
                    for (int f = 0; f < 4; ++f)
                        fb[f] = *reinterpret_cast<const bf16x8_t *>(pb + rowoff[f] + ct[f][kx]);
#pragma unroll
                    for (int i = 0; i < TCO; ++i)
                        fa[i] = *reinterpret_cast<const bf16x8_t *>(ws + (fq * BCO + i * 16 + fj) * 16);
                    // own pieces of slice s+1 (and older) have landed; right after an epilogue its stores are younger
                    // than that group and may stay in flight
                    if (relaxed > 0) {
                        wait_vmcnt<4 + N_ST>();
                        --relaxed;
                    } else {
                        wait_vmcnt<4>();
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers (R2)
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- M(s)
#pragma unroll
                    for (int i = 0; i < TCO; ++i)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[f], acc[i][f], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        stream_epilogue<BCO, TW, EPI, 4, EPI == SEPI_CHAIN, true, EPI == SEPI_CHAIN>(a, acc, co_tile, n, y0, x0, frow, fj, fq, s_chain, nullptr, 0, a.x4 != 0);
        if (!has_next) break;
        tile = next;
        n = nn;
        y0 = ny0;
        x0 = nx0;
        relaxed = 2;   // pd[] holds this tile's chunk-0 descriptors (its fill was issued during the previous tile's last chunk)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the offset barrier of group 1
}

int v2x_conv_stream_pc_launch(const StreamArgs &a, hipStream_t s);   // conv_stream_pc.hip

int v2x_num_cus() {   // also used by conv_halo_pair.hip
    // hipDeviceGetAttribute, NOT hipGetDeviceProperties: one call of the latter anywhere in the process made EVERY kernel of
    // the step 4-8 % slower on the MI355X boxes (interleaved A/B, 4 250 vs 4 440 frames/s; a clock/power-state side effect)
    static int n[V2X_MAX_DEVICES] = {0};   // per device, like the attribute caches
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= V2X_MAX_DEVICES) return 256;
    if (n[dev] == 0)
        n[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    return n[dev];
}

// ---- 8-wave form with THREE taps per synchronisation ("stream8g") ------------------------------------------------------------
// tools/stream8_phase_probe.sh (one phase compiled out at a time, dominant kernel at 320 maps): full step 550 us; without the
// MFMAs 519; without the DMAs 414; without the fragment reads 410; with neither 382 -- the 8-wave kernel above is bound by its
// LOAD phase (per step and wave: 2 LDS-DMA instructions, 12 ds_read_b128, address selects, a counted wait and two barriers
// ~ 825 cycles) against 512 cycles of MFMAs.  Here a step is a TAP COLUMN of a 32-channel chunk -- the three taps (ky = 0..2) of
// one kx -- so a synchronisation interval carries 96 MFMAs per wave instead of 32:
//   * pixel operands are shared between the tap rows: the wave's two output rows need 4 patch rows (3 at half resolution)
//     -> 8 B fragments per step instead of 12, read in the load phase;
//   * weight operands stream through registers in HALF taps (TCO/2 fragments): the reads of half h+1 are issued before the
//     MFMAs of half h, inside the MFMA phase (two alternating sets; a whole tap would not fit beside 128 accumulators);
//   * the weight ring holds 3 steps of 3 slices (72 KiB at 128 rows).  Group 0's waves DMA their pieces of step s+1, group 1's
//     waves theirs of step s+2, each in its own load phase L(s); every wave drains its own DMAs (vmcnt(0)) at the end of its
//     MFMA phase, one full phase later.  With the half-step offset of the two groups this is hazard-free with THREE slots:
//     slot (s+1)%3 is written by group 0 in interval 2s, its old content (step s-2) was last read by group 1 in interval 2s-2;
//     slot (s+2)%3 is written by group 1 in interval 2s+1, its old content (step s-1) was last read by group 0 in interval
//     2s-1 and by group 1 itself in interval 2s.  Pieces of step s+1 are drained by interval 2s+1 (group 0) / 2s (group 1),
//     first read in interval 2s+2.  The next chunk's patch is DMA'd during the steps kx = 0, 1 (drained by interval 6kc+4, first
//     read in 6kc+6) into the buffer last read in the load phases of chunk kc-1 (interval 6kc-1 at the latest).
//   No counted waits, no dummy DMAs.  Persistent over tiles like the kernel above (the ring and the patch fill wrap into the
//   next tile).  K order is (chunk, kx, ky): the fp32 sums differ from the other streamed kernels in their last bits, so this
//   form replaces them for a layer everywhere or nowhere (the choice depends on the layer's shape only).
// the load phase's DMA issue of conv3x3_stream8g_kernel (a macro, not a lambda: captured by reference, `pd` and `nw` went to scratch)
#define V2X_STREAM8G_ISSUE_DMAS \
                if (grp == 1) { \
                    const int ahead = st + 2; \
                    const int wslot = slot == 0 ? 2 : slot - 1; \
                    if (ahead < S3) nw = issue_weights(ahead, wslot, wv, 4); \
                    else if (has_next && ahead - S3 < S3) nw = issue_weights(ahead - S3, wslot, wv, 4); \
                } else if (fill && kx < 2) { \
                    const int npieces = hfn ? (PH0 * PW0 * 4 + 63) / 64 : PATCH8_PIECES; \
_Pragma("unroll") \
                    for (int t = 0; t < 5; ++t) { \
                        const int tt = kx * 5 + t; \
                        if (wv + 4 * tt >= npieces) break; \
                        int d = pd[0]; \
_Pragma("unroll") \
                        for (int u = 1; u < 10; ++u) d = (tt == u) ? pd[u] : d; \
                        issue_piece(d, kcn, tt, (gc + 1) & 1); \
                    } \
                }
//   WT (wave tiling): false -- a wave owns ALL BCO channels x 64 pixels (2 rows) of its group's 8x32 pixels: 24 weight + 8 pixel fragment
//   reads per step; true -- a wave owns HALF the channels x 128 pixels (4 rows): 12 + 12 reads per step for the same 96 MFMAs (-25 % LDS
//   reads: -3.1...3.8 % per layer in the paired A/B).  Same K order, bit-identical results.
template <int BCO, int EPI, bool WT = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_stream8g_kernel(const StreamArgs a) {
    constexpr int TH = 16, TW = 32;
    constexpr int PW = TW + 2, PH = TH + 2, PW0 = TW / 2 + 2, PH0 = TH / 2 + 2;
    constexpr int TCO = BCO / 16, HCO = TCO / 2;            // channel tiles per tap / per half tap
    constexpr int W_PIECES = BCO / 16;                     // 1-KiB pieces per tap slice
    constexpr int SLICE_BYTES = BCO * 64;
    constexpr int STEP_BYTES = 3 * SLICE_BYTES;            // one tap column of a chunk
    constexpr int NWD = (3 * W_PIECES + 3) / 4;            // weight DMAs per wave of GROUP 1 and step (6 at 128 rows, 5 at 96)
    static_assert(EPI != SEPI_CHAIN, "the chained epilogue stays on conv3x3_stream8_kernel (its operands need the LDS; from L2: +5.3 %)");
    static_assert(TCO % 2 == 0 && PH * PW * 4 <= (PATCH8_PIECES - 1) * 64, "half taps; patch fits its buffer");
    constexpr int AT = WT ? HCO : TCO;                     // accumulator tiles of a wave: channel tiles x pixel fragments
    constexpr int NF = WT ? 8 : 4;
    constexpr int NB = WT ? 12 : 8;                        // pixel fragments a wave reads per step
    constexpr int BT = (HCO % 2 == 0) ? 2 : HCO;           // WT: channel tiles per MFMA block (2 -> blocks of 16 MFMAs, 3 -> of 24)
    constexpr int PSH = V2X_STREAM_PSWZ_BUILD;             // patch swizzle shift
    // output-store instructions per wave and tile (the counted waits after an epilogue leave exactly these in flight): 8-byte stores per
    // (16-channel tile, 16-pixel fragment)
    constexpr int N_ST = (EPI == SEPI_GRU) ? (TCO / 3) * 2 : TCO * 2;   // dwordx4 stores (stream_epilogue X4); on the dwordx2 path (unaligned rows) the counted wait is merely stricter

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_ring = smem;                                   // 3 x STEP_BYTES
    char *s_patch = smem + 3 * STEP_BYTES;                 // 2 x PATCH8_BYTES

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int grp = wave >> 2, wv = wave & 3;
    const int fj = lane & 15, fq = lane >> 4;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
        if (a.xcd_walk != 0 && a.n_co_tiles == 8 && r == 0 && q == 32) {
            // an XCD's 32 workgroups of a round: 8 pixel tiles x 4 channel tiles (5.0 MB of patches + 3.5 MB of weights through its 4-MiB L2)
            // instead of 4 x 8 (2.5 + 7.1 MB); the two XCDs of a pair share the 8 pixel tiles.  Measured on the ConvGRU: FETCH_SIZE 1.47 -> 1.24 GB
            // per 320 maps, time -0.1 +- 0.1 %, bit-identical (tools/gru_xcd_walk_probe.sh)
            const int x = bid >> 5, i = bid & 31;
            bid = (((x >> 1) * 8 + (i >> 2)) << 3) + (x & 1) * 4 + (i & 3);
        }
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    const int co_tile = bid % a.n_co_tiles;
    const int txy = a.tiles_x * a.tiles_y;
    const int nc0 = a.C0 >> 5, nchunks = (a.C0 + a.C1) >> 5;
    const int S3 = nchunks * 3;                            // steps (tap columns) per tile
    const uint16_t *wbase = a.w + (size_t)co_tile * nchunks * 9 * (BCO * 32);
    const void *zero_page = a.w + (size_t)a.n_co_tiles * nchunks * 9 * (BCO * 32);

    auto tile_coords = [&](int t, int &n, int &y0, int &x0) {
        const int px_tile = t / a.n_co_tiles;
        n = px_tile / txy;
        const int trem = px_tile - n * txy;
        const int ty = trem / a.tiles_x;
        y0 = ty * TH;
        x0 = (trem - ty * a.tiles_x) * TW;
    };
    // lane id from VOLATILE asm: values derived from it cannot be hoisted out of the tile / chunk / step loops (as lane constants
    // they would be -- dozens of registers kept, and spilled, across the MFMA phases)
    auto fresh_lane = [&]() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    auto desc = [&](int n, int y0, int x0, int t, bool hf) -> int {   // DMA source descriptor of patch piece (wv + 4t) (group 0 fills the patch)
        const int L = (wv + 4 * t) * 64 + fresh_lane();
        const int pix = L >> 2, phys = L & 3;
        if (!hf) {
            const int pr = pix / PW, pc = pix - pr * PW;
            const int y = y0 - 1 + pr, x = x0 - 1 + pc;
            const bool ok = pix < PH * PW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
            return ok ? ((((n * a.H + y) * a.W + x) << 5) | ((phys ^ ((pc >> PSH) & 3)) << 3)) : -1;
        }
        const int Hs = a.H >> 1, Ws = a.W >> 1;
        const int pr = pix / PW0, pc = pix - pr * PW0;
        const int y = (y0 >> 1) - 1 + pr, x = (x0 >> 1) - 1 + pc;
        const bool ok = pix < PH0 * PW0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
        return ok ? ((((n * Hs + y) * Ws + x) << 5) | ((phys ^ ((pc >> PSH) & 3)) << 3)) : -1;
    };
    auto issue_piece = [&](int d, int kc, int t, int buf) {
        const bool first = kc < nc0;
        const uint16_t *src = first ? a.in0 : a.in1;
        const unsigned cs = first ? (unsigned)a.C0 : (unsigned)a.C1;
        const unsigned off = (unsigned)(d >> 5) * cs + (unsigned)((first ? kc : kc - nc0) * 32 + (d & 31));
        glds16s_aux<V2X_STREAM8G_PATCH_AUX>(d >= 0 ? (const void *)(src + off) : zero_page, s_patch + buf * PATCH8_BYTES + (wv + 4 * t) * 1024);
    };
    // weight pieces of step `st` (chunk st / 3, tap column st % 3): piece p = wq + NQ * u, p < 3 * W_PIECES, for worker wq of NQ
    // (the prologue spreads a step over all 8 waves, the steady state over the 4 waves of group 1);
    // piece p = (tap row ky = p / W_PIECES, 1-KiB piece p % W_PIECES of that tap's slice)
    auto issue_weights = [&](int st, int slot, int wq, int NQ) -> int {   // -> number of DMAs issued (wave-uniform)
        const int kc = st / 3, kx = st - kc * 3;
        const int lane_w = fresh_lane();
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < NWD; ++u) {
            const int p = wq + NQ * u;
            if (p >= 3 * W_PIECES) break;                  // wave-uniform (96-row tiles: 18 pieces)
            ++cnt;
            const int ky = p / W_PIECES, pis = p - ky * W_PIECES;
            glds16s_aux<V2X_STREAM8G_WEIGHT_AUX>(wbase + (size_t)(kc * 9 + ky * 3 + kx) * (BCO * 32) + pis * 512 + lane_w * 8,
                    s_ring + slot * STEP_BYTES + ky * SLICE_BYTES + pis * 1024);
        }
        return cnt;
    };
    // wait until at most `keep` vector-memory operations are outstanding, keep in {0, NWD-1, NWD} (+ N_ST): the immediate must be
    // a constant, the choice is wave-uniform
    auto wait_keep = [&](int keep, bool plus_stores) {
        if (plus_stores) {
            if (keep == NWD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWD + N_ST) : "memory");
            else if (keep == NWD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWD - 1 + N_ST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_ST) : "memory");
        } else {
            if (keep == NWD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWD) : "memory");
            else if (keep == NWD - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWD - 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };

    const int coh = WT ? (wv & 1) : 0;                     // WT: the wave's channel half
    const int R0 = WT ? 8 * grp + 4 * (wv >> 1) : 8 * grp + 2 * wv;   // the wave's first output row (even)
    int frow[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) frow[f] = R0 + (f >> 1);
    const bool chunk0_half = (nc0 > 0) && a.up0;

    int tile = bid;
    int n, y0, x0;
    tile_coords(tile, n, y0, x0);
    // Division of labour in the steady state: GROUP 0 fills the next chunk's patch (10 pieces per wave: t = 0..4 at kx = 0, 5..9 at
    // kx = 1, drained at the end of its load phase of kx = 2), GROUP 1 streams the weights: ALL pieces of step s+2 in its load phase
    // L(s) (interval 2s+1) into slot (s+2)%3 -- last read by group 0 in interval 2s-1 and by group 1 itself in interval 2s -- and
    // waits for them at the end of its NEXT load phase L(s+1) (interval 2s+3; counted: the NWD newest DMAs are that phase's own),
    // one interval before group 0 reads them.  A weight DMA thus has an MFMA phase plus a load phase (~1.1 us) to land, a patch
    // DMA two to four intervals; the first form of this kernel drained every wave's DMAs after one MFMA phase (0.7 us) and the
    // 128-row layers gained nothing.
    int pd[10];
    bool pd_half = chunk0_half;
    if (grp == 0) {
#pragma unroll
        for (int t = 0; t < 10; ++t) pd[t] = desc(n, y0, x0, t, pd_half);
#pragma unroll
        for (int t = 0; t < 10; ++t) issue_piece(pd[t], 0, t, 0);
    }
    // prologue: patch of chunk 0 (group 0), weights of steps 0 and 1 (all 8 waves share them)
    {
        constexpr int NP8 = (3 * W_PIECES + 7) / 8;
        for (int stp = 0; stp < (S3 > 1 ? 2 : 1); ++stp) {
            const int lane_w = fresh_lane();
#pragma unroll
            for (int u = 0; u < NP8; ++u) {
                const int p = wave + 8 * u;
                if (p >= 3 * W_PIECES) break;
                const int ky = p / W_PIECES, pis = p - ky * W_PIECES;
                glds16s_aux<V2X_STREAM8G_WEIGHT_AUX>(wbase + (size_t)((stp / 3) * 9 + ky * 3 + (stp % 3)) * (BCO * 32) + pis * 512 + lane_w * 8,
                        s_ring + stp * STEP_BYTES + ky * SLICE_BYTES + pis * 1024);
            }
        }
    }
    // epilogue parameters of this workgroup's channel tile -> LDS (1 KiB behind the patches): see stream_epilogue
    float *s_ss = reinterpret_cast<float *>(smem + 3 * STEP_BYTES + 2 * PATCH8_BYTES);
    // (paired A/B: plain layers -0.7 ... -3.4 %, most on the 12-step layers; the ConvGRU +0.7 % -- its table stays in global memory)
    if constexpr (V2X_STREAM_LSS_BUILD != 0 && EPI != SEPI_GRU) {
        if constexpr (EPI == SEPI_GRU) {
            constexpr int NH = BCO / 3;                    // hidden channels of the tile
            for (int i = tid; i < NH * 4; i += 512) {
                const int hc = co_tile * NH + (i >> 2);
                s_ss[i] = hc < a.Cout ? a.scale[(size_t)hc * 4 + (i & 3)] : 0.f;
            }
        } else {
            for (int i = tid; i < BCO; i += 512) {
                const int co = co_tile * BCO + i;
                s_ss[i] = co < a.Cout ? a.scale[co] : 0.f;
                s_ss[BCO + i] = co < a.Cout ? a.shift[co] : 0.f;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();            // half-step offset

    int g = 0;     // global step counter
    bool relaxed = false;
    int slot = 0;  // ring slot of the current step = g % 3
    int gc = 0;    // global chunk counter: patch buffer = gc & 1
    for (;;) {
        const int next = tile + nwg;
        const bool has_next = next < n_tiles;
        int nn = 0, ny0 = 0, nx0 = 0;
        if (has_next) tile_coords(next, nn, ny0, nx0);

        f32x4_t acc[AT][NF];
#pragma unroll
        for (int i = 0; i < AT; ++i)
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[i][f] = (f32x4_t)(0.f);

        for (int kc = 0; kc < nchunks; ++kc, ++gc) {
            const char *pb = s_patch + (gc & 1) * PATCH8_BYTES;
            const bool half = (kc < nc0) && a.up0;
            const bool last_chunk = kc + 1 == nchunks;
            const int kcn = last_chunk ? 0 : kc + 1;
            const bool fill = !last_chunk || has_next;
            const bool hfn = (kcn < nc0) && a.up0;
            if (grp == 0 && ((last_chunk && has_next) || (fill && hfn != pd_half))) {
                const int dn = last_chunk ? nn : n, dy = last_chunk ? ny0 : y0, dx = last_chunk ? nx0 : x0;
#pragma unroll
                for (int t = 0; t < 10; ++t) {
                    pd[t] = desc(dn, dy, dx, t, hfn);
                    __builtin_amdgcn_sched_barrier(0);
                }
                pd_half = hfn;
            }
            // one chunk = three steps (tap columns).  ONE code path for both source resolutions: the pixel fragment a (row r, tap row
            // ky) pair uses is always B[r + ky]; at half resolution (nearest x2 upsample) the four entries are loaded from the patch
            // rows {0, 1, 1, 2} -- the resolution only enters address arithmetic (a run-time choice between REGISTERS would be selects
            // on 32 of them, and two specialised copies of this body made the allocator shuffle accumulators through scratch).
            const int sh = half ? 1 : 0;
            const int row_bytes = (half ? PW0 : PW) * 64;
            const int row0 = (R0 >> sh) * row_bytes;            // the wave's first patch row
#pragma unroll 1   // rolled: unrolled copies of this body hoist ~100 lane constants and spill
            for (int kx = 0; kx < 3; ++kx, ++g, slot = (slot == 2 ? 0 : slot + 1)) {
                const int st = kc * 3 + kx;
                const int ln = fresh_lane();
                const int fjl = ln & 15, fql = ln >> 4;
                // ---- L: group 1 streams the weights of step st+2 (wrapping into the next tile), group 0 the next chunk's patch
                int nw = 0;   // weight DMAs this wave issues in this phase
                V2X_STREAM8G_ISSUE_DMAS
                __builtin_amdgcn_sched_barrier(0);
                // pixel fragments of the tap column: B[q * 2 + ch], q = r + ky = 0..3
                bf16x8_t B[NB];
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const int col = ch * 16 + fjl + kx;
                    const int pc = ((col - sh) >> sh) + sh;     // full: col + kx;  half: ((col + kx - 1) >> 1) + 1
                    const int coff = ((pc << 2) + (fql ^ ((pc >> PSH) & 3))) * 16;
#pragma unroll
                    for (int q = 0; q < NB / 2; ++q) {
                        B[q * 2 + ch] = *reinterpret_cast<const bf16x8_t *>(pb + row0 + (((q - sh) >> sh) + sh) * row_bytes + coff);
                    }
                }
                const char *ws = s_ring + slot * STEP_BYTES + (fql * BCO + fjl) * 16 + coh * (HCO * 256);   // + compile-time offsets below
                bf16x8_t A[2][WT ? BT : HCO];                    // two alternating sets: the next block's fragments land under this block's MFMAs
#pragma unroll
                for (int i = 0; i < (WT ? BT : HCO); ++i) A[0][i] = *reinterpret_cast<const bf16x8_t *>(ws + i * 256);
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): fragments in registers before the patch / ring may be overwritten
                // drain: group 1 -- the weights issued one step ago (everything but this phase's own NWD DMAs; right after an epilogue
                // the tile's output stores are younger than those and may stay in flight); group 0 -- at kx = 2, the patch it issued
                // during kx = 0, 1 (two to four intervals ago), together with any older stores
                if (grp == 1) {
                    wait_keep(nw, relaxed);
                    relaxed = false;
                } else if (kx == 2) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- M: six half taps.  The weight fragments of half h+1 are read in the MIDDLE of half h's MFMA block: the compiler
                // waits for them with lgkmcnt(0) before their first use (it does not count LDS reads individually here), so reads
                // issued right before a block would be waited for at once -- six exposed LDS latencies per step (the first form:
                // MFMA phases alone 458 us against 318 us of MFMA work).  Issued after the first part of the block they land under
                // its second part.
                if constexpr (WT) {
                    // blocks of BT channel tiles x 8 pixel fragments.  The next block's BT weight fragments are read after the
                    // first tile's 8 MFMAs and land under the rest of the block (the compiler waits lgkmcnt(0) before their first use).
                    constexpr int NBLK = 3 * HCO / BT;
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) {
                        const int ky = (b * BT) / HCO, i0 = (b * BT) % HCO;
                        auto mma_tiles = [&](int j0, int j1) __attribute__((always_inline)) {
#pragma unroll
                            for (int j = j0; j < j1; ++j)
#pragma unroll
                                for (int f = 0; f < 8; ++f)
                                    acc[i0 + j][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[b & 1][j], B[((f >> 1) + ky) * 2 + (f & 1)], acc[i0 + j][f], 0, 0, 0);
                        };
                        __builtin_amdgcn_sched_barrier(0);
                        mma_tiles(0, 1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (b + 1 < NBLK) {
                            const int ky1 = ((b + 1) * BT) / HCO, j1 = ((b + 1) * BT) % HCO;
#pragma unroll
                            for (int j = 0; j < BT; ++j) A[(b + 1) & 1][j] = *reinterpret_cast<const bf16x8_t *>(ws + ky1 * SLICE_BYTES + (j1 + j) * 256);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        mma_tiles(1, BT);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                constexpr int H1 = 1;   // channel tiles of the first part (the prefetch point inside a half tap: moving it measured 0.0 %)
#pragma unroll
                for (int h = 0; h < 6; ++h) {
                    const int ky = h >> 1, hh = h & 1;
                    auto mma_part = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
                        for (int i = i0; i < i1; ++i)
#pragma unroll
                            for (int f = 0; f < 4; ++f)
                                acc[hh * HCO + i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[h & 1][i], B[((f >> 1) + ky) * 2 + (f & 1)], acc[hh * HCO + i][f], 0, 0, 0);
                    };
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(0, H1);
                    __builtin_amdgcn_sched_barrier(0);
                    if (h + 1 < 6) {
                        const int ky1 = (h + 1) >> 1, hh1 = (h + 1) & 1;
#pragma unroll
                        for (int i = 0; i < HCO; ++i) A[(h + 1) & 1][i] = *reinterpret_cast<const bf16x8_t *>(ws + ky1 * SLICE_BYTES + (hh1 * HCO + i) * 256);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    mma_part(H1, HCO);
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        constexpr bool LSS = V2X_STREAM_LSS_BUILD != 0 && EPI != SEPI_GRU;
        // (the GRU's table is float4 per hidden channel: a channel half of the tile starts BCO / 6 channels = 4 * BCO / 6 floats in)
        if constexpr (WT)
            stream_epilogue<BCO / 2, TW, EPI, 8, false, true, true>(a, acc, co_tile * 2 + coh, n, y0, x0, frow, fj, fq, nullptr,
                                                                    LSS ? (lds_cf_t *)s_ss + coh * (EPI == SEPI_GRU ? 4 * (BCO / 6) : BCO / 2) : (lds_cf_t *)nullptr, BCO, a.x4 != 0);
        else stream_epilogue<BCO, TW, EPI, 4, false, true, true>(a, acc, co_tile, n, y0, x0, frow, fj, fq, nullptr, LSS ? (lds_cf_t *)s_ss : (lds_cf_t *)nullptr, BCO, a.x4 != 0);
        if (!has_next) break;
        tile = next;
        n = nn;
        y0 = ny0;
        x0 = nx0;
        relaxed = true;   // the stores just issued are younger than the weight DMAs the next load phase waits for
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();            // balance the offset barrier of group 1
}

template <int BCO, int EPI, bool WT = false>
static int launch_stream8g(const StreamArgs &a, hipStream_t s) {
    constexpr int smem = 3 * 3 * BCO * 64 + 2 * PATCH8_BYTES + 1024;   // 153 KiB at 128 rows, 135 KiB at 96 (+1 KiB: epilogue parameters)
    static_assert(smem <= 160 * 1024, "LDS budget");
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_stream8g_kernel<BCO, EPI, WT>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    int grid = n_tiles;
    int g = v2x_num_cus() / a.n_co_tiles * a.n_co_tiles;   // persistent: a workgroup's tiles share one channel tile
    if (g > 0 && g < n_tiles) grid = g;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_stream8g_kernel");
    return V2X_OK;
}

template <int BCO, int EPI>
static int launch_stream8(const StreamArgs &a, hipStream_t s) {
    constexpr int smem = RING * BCO * 64 + 2 * PATCH8_BYTES + (EPI == SEPI_CHAIN ? chain_lds_bytes<BCO>() : 0);  // 112 KiB at BCO=128 (+33 chained): one 8-wave workgroup per CU
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_stream8_kernel<BCO, EPI>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    // persistent grid: one workgroup per CU, rounded down to a multiple of n_co_tiles so that a workgroup's tiles
    // (bid, bid + grid, ...) all belong to one channel tile; fewer tiles than CUs (or a channel-tile count that does not
    // divide) -> one tile per workgroup.  V2X_STREAM_PERSIST=0 forces the one-tile form (A/B runs).
    const int n_tiles = a.n_px_tiles * a.n_co_tiles;
    int grid = n_tiles;
    if (v2x_tune(V2X_TUNE_STREAM_PERSIST) != 0) {
        int g = v2x_num_cus() / a.n_co_tiles * a.n_co_tiles;
        if (EPI != SEPI_CHAIN && g > 0 && g < n_tiles && ((a.C0 + a.C1) >> 5) >= 2) grid = g;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_stream8_kernel");
    return V2X_OK;
}

// ---- "wide" 4-wave form for the 64-row layers ----------------------------------------------------------
// conv7_1 (192 -> 64), conv1_2 -> conv3d_1 and conv7_2 (64 -> 64) at 128x128 have only ONE 64-row channel tile: with the
// 256-pixel tile above a wave does 16 MFMAs per step against 1 weight + 1 patch DMA and a barrier (twice the overhead per
// MFMA of the 128-row tiles), and every 256 pixels re-stream the layer's whole weight tensor from L2 (4.5 GB per conv7_1
// launch).  Here a wave owns 128 pixels (4 rows x 32 columns, 8 B fragments): 32 MFMAs per step from 4 + 8 fragment
// reads, the same ratios as the 128-row kernels, and a weight slice is fetched once per 512 pixels.  The 16x32 tile's
// patch is 40 KiB, so it is SINGLE-buffered (ring 16 + patch 40 = 56 KiB -> two workgroups per CU, as in
// conv_stream_s2.hip): at a chunk boundary everything drains, the patch is refilled and the co-resident workgroup's MFMAs
// cover the refill.  Weight ring, counted waits and K order are those of the 256-pixel kernel.
constexpr int WPATCH_PIECES = 39;               // 18 x 34 pixels x 64 B = 38.25 KiB
constexpr int WPATCH_BYTES = WPATCH_PIECES * 1024;

template <int BCO, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wide_kernel(const StreamArgs a) {
    constexpr int TH = 16, TW = 32, NF = 8;
    constexpr int PW = TW + 2, PH = TH + 2, PW0 = TW / 2 + 2, PH0 = TH / 2 + 2;
    constexpr int TCO = BCO / 16;
    constexpr int SLICE_BYTES = BCO * 64;
    constexpr int PPW = (WPATCH_PIECES + 3) / 4;   // patch pieces per wave (10)
    static_assert(BCO == 64, "wide form: one 64-row channel tile (1 weight DMA per wave and step)");
    static_assert(PH * PW * 4 <= WPATCH_PIECES * 64, "patch must fit its buffer");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_ring = smem;
    char *s_patch = smem + RING * SLICE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int co_tile = bid % a.n_co_tiles;
    const int px_tile = bid / a.n_co_tiles;
    const int txy = a.tiles_x * a.tiles_y;
    const int n = px_tile / txy;
    const int trem = px_tile - n * txy;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    const int nc0 = a.C0 >> 5, nchunks = (a.C0 + a.C1) >> 5;
    const int S = nchunks * 9;
    const uint16_t *wbase = a.w + (size_t)co_tile * nchunks * 9 * (BCO * 32);
    const void *zero_page = a.w + (size_t)a.n_co_tiles * nchunks * 9 * (BCO * 32);
    char *s_chain = smem + RING * SLICE_BYTES + WPATCH_BYTES;
    if constexpr (EPI == SEPI_CHAIN) stage_chain<BCO, 256>(a, s_chain);

    // DMA descriptors of this wave's patch pieces (piece = wave + 4t) for the resolution of the chunk being filled;
    // rebuilt when the resolution changes (at most twice per tile)
    int pd[PPW];
    auto build_pd = [&](bool hf) {
        // opaque copy of the lane id: without it the compiler evaluates BOTH resolutions' descriptors for all ten pieces
        // ahead of the chunk loop (loop-invariant code motion) and keeps ~40 intermediates alive across it (spills)
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const int L = (wave + 4 * t) * 64 + lane_o;
            const int pix = L >> 2, phys = L & 3;
            int d;
            if (!hf) {
                const int pr = pix / PW, pc = pix - pr * PW;
                const int y = y0 - 1 + pr, x = x0 - 1 + pc;
                const bool ok = pix < PH * PW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                d = ok ? ((((n * a.H + y) * a.W + x) << 5) | ((phys ^ PSWZ(pc)) << 3)) : -1;
            } else {
                const int Hs = a.H >> 1, Ws = a.W >> 1;
                const int pr = pix / PW0, pc = pix - pr * PW0;
                const int y = (y0 >> 1) - 1 + pr, x = (x0 >> 1) - 1 + pc;
                const bool ok = pix < PH0 * PW0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
                d = ok ? ((((n * Hs + y) * Ws + x) << 5) | ((phys ^ PSWZ(pc)) << 3)) : -1;
            }
            pd[t] = d;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // column offsets depend on the fragment's column half (f & 1) only: 6 registers instead of 24
    int frow[NF], ct[2][3];
#pragma unroll
    for (int f = 0; f < NF; ++f) frow[f] = 4 * wave + (f >> 1);
    auto build_ct = [&](bool hf) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = c * 16 + fj;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int pc = hf ? (((col + kx - 1) >> 1) + 1) : (col + kx);
                ct[c][kx] = ((pc << 2) + (fq ^ PSWZ(pc))) * 16;
            }
        }
    };

    const uint16_t *wsrc = wbase + lane * 8 + wave * 512;
    auto issue_weights = [&](int s) {   // one DMA (BCO = 64: 4 pieces, one per wave)
        glds16s(wsrc + (size_t)s * (BCO * 32), s_ring + (s & (RING - 1)) * SLICE_BYTES + wave * 1024);
    };
    auto issue_patch = [&](int kc, bool hf) {
        const bool first = kc < nc0;
        const uint16_t *src = first ? a.in0 : a.in1;
        const unsigned cs = first ? (unsigned)a.C0 : (unsigned)a.C1;
        const int npieces = hf ? (PH0 * PW0 * 4 + 63) / 64 : WPATCH_PIECES;   // the half-resolution patch is 3x smaller
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            if (wave + 4 * t >= npieces) break;              // wave-uniform
            int d = pd[t];
            asm volatile("" : "+v"(d));   // keep the address arithmetic HERE: hoisted out of the chunk loop it becomes ten
                                          // live 64-bit addresses on top of 176 busy registers (it spilled)
            const unsigned off = (unsigned)(d >> 5) * cs + (unsigned)((first ? kc : kc - nc0) * 32 + (d & 31));
            glds16s(d >= 0 ? (const void *)(src + off) : zero_page, s_patch + (wave + 4 * t) * 1024);
        }
    };

    f32x4_t acc[TCO][NF];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    issue_weights(0);   // S >= 18
    issue_weights(1);
    issue_weights(2);

    int s = 0;
    bool cur_half = false;
    for (int kc = 0; kc < nchunks; ++kc) {
        const bool half = (kc < nc0) && a.up0;
        const int sh = half ? 1 : 0, row_bytes = (half ? PW0 : PW) * 64;
        if (kc == 0 || half != cur_half) {   // wave-uniform
            build_pd(half);
            build_ct(half);
            cur_half = half;
        }
        // ---- chunk boundary: everyone has left the previous chunk's patch (and ring slot s-1) -> refill, drain, meet
        if (kc > 0) __builtin_amdgcn_s_barrier();
        issue_patch(kc, half);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int ky = 0; ky < 3; ++ky) {
            int rowoff[NF / 2];          // fragments 2r, 2r+1 share a row
#pragma unroll
            for (int r = 0; r < NF / 2; ++r) rowoff[r] = (((4 * wave + r + ky - sh) >> sh) + sh) * row_bytes;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx, ++s) {
                const int tap = ky * 3 + kx;
                if (tap > 0) {
                    // this step's slice has landed (groups s+1, s+2 may be in flight); everyone is done with ring slot s-1
                    if (s + 2 >= S) wait_vmcnt<0>();
                    else wait_vmcnt<2>();
                    __builtin_amdgcn_s_barrier();
                }
                if (s + 3 < S) issue_weights(s + 3);
                const char *ws = s_ring + (s & (RING - 1)) * SLICE_BYTES;
                bf16x8_t fa[TCO], fb[NF];
#pragma unroll
                for (int f = 0; f < NF; ++f) fb[f] = *reinterpret_cast<const bf16x8_t *>(s_patch + rowoff[f >> 1] + ct[f & 1][kx]);
#pragma unroll
                for (int i = 0; i < TCO; ++i)
                    fa[i] = *reinterpret_cast<const bf16x8_t *>(ws + (fq * BCO + i * 16 + fj) * 16);
#pragma unroll
                for (int i = 0; i < TCO; ++i)
#pragma unroll
                    for (int f = 0; f < NF; ++f)
                        acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[f], acc[i][f], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, TCO + NF, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TCO * NF, 0);
            }
        }
    }
    stream_epilogue<BCO, TW, EPI, NF, EPI == SEPI_CHAIN, true, true>(a, acc, co_tile, n, y0, x0, frow, fj, fq, s_chain, nullptr, 0, a.x4 != 0);
}

// ---- "wide3": the wide form with THREE taps per synchronisation (round 2) -------------------------------------------
// The step of the kernel above is one tap: 32 MFMAs from 4 weight + 8 pixel fragment reads, one counted wait and one barrier.  Here a
// step is one tap COLUMN kx of a chunk (the three taps ky = 0..2), as in conv3x3_stream8g_kernel: the 12 pixel fragments of the wave's
// six patch rows serve all three taps (fragment (row r, tap ky) is row r + ky), so a step is 96 MFMAs from 12 + 12 fragment reads, ONE
// counted wait and ONE barrier -- a third of the synchronisations and two thirds of the LDS reads per MFMA.  The weight ring holds
// 3 steps of 3 slices (36 KiB); with the single-buffered 39-KiB patch that is 75 KiB: still two workgroups per CU.  K order is
// (chunk, kx, ky) like stream8g: results agree with the other streamed kernels to one bf16 rounding, not bitwise.
template <int BCO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv3x3_wide3_kernel(const StreamArgs a) {
    constexpr int TH = 16, TW = 32, NF = 8, NB = 12;
    constexpr int PW = TW + 2, PH = TH + 2, PW0 = TW / 2 + 2, PH0 = TH / 2 + 2;
    constexpr int TCO = BCO / 16;
    constexpr int SLICE_BYTES = BCO * 64, STEP_BYTES = 3 * SLICE_BYTES;
    constexpr int PPW = (WPATCH_PIECES + 3) / 4;   // patch pieces per wave (10)
    static_assert(BCO == 64, "wide form: one 64-row channel tile (one weight piece per wave and tap)");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *s_ring = smem;                           // 3 x STEP_BYTES
    char *s_patch = smem + 3 * STEP_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fj = lane & 15, fq = lane >> 4;

    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int co_tile = bid % a.n_co_tiles;
    const int px_tile = bid / a.n_co_tiles;
    const int txy = a.tiles_x * a.tiles_y;
    const int n = px_tile / txy;
    const int trem = px_tile - n * txy;
    const int ty = trem / a.tiles_x;
    const int tx = trem - ty * a.tiles_x;
    const int y0 = ty * TH, x0 = tx * TW;

    const int nc0 = a.C0 >> 5, nchunks = (a.C0 + a.C1) >> 5;
    const int S3 = nchunks * 3;
    const uint16_t *wbase = a.w + (size_t)co_tile * nchunks * 9 * (BCO * 32);
    const void *zero_page = a.w + (size_t)a.n_co_tiles * nchunks * 9 * (BCO * 32);

    int pd[PPW];
    auto build_pd = [&](bool hf) {
        int lane_o = lane;   // opaque: see conv3x3_wide_kernel
        asm volatile("" : "+v"(lane_o));
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            const int L = (wave + 4 * t) * 64 + lane_o;
            const int pix = L >> 2, phys = L & 3;
            int d;
            if (!hf) {
                const int pr = pix / PW, pc = pix - pr * PW;
                const int y = y0 - 1 + pr, x = x0 - 1 + pc;
                const bool ok = pix < PH * PW && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
                d = ok ? ((((n * a.H + y) * a.W + x) << 5) | ((phys ^ PSWZ(pc)) << 3)) : -1;
            } else {
                const int Hs = a.H >> 1, Ws = a.W >> 1;
                const int pr = pix / PW0, pc = pix - pr * PW0;
                const int y = (y0 >> 1) - 1 + pr, x = (x0 >> 1) - 1 + pc;
                const bool ok = pix < PH0 * PW0 && (unsigned)y < (unsigned)Hs && (unsigned)x < (unsigned)Ws;
                d = ok ? ((((n * Hs + y) * Ws + x) << 5) | ((phys ^ PSWZ(pc)) << 3)) : -1;
            }
            pd[t] = d;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int frow[NF], ct[2][3];
#pragma unroll
    for (int f = 0; f < NF; ++f) frow[f] = 4 * wave + (f >> 1);
    auto build_ct = [&](bool hf) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = c * 16 + fj;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int pc = hf ? (((col + kx - 1) >> 1) + 1) : (col + kx);
                ct[c][kx] = ((pc << 2) + (fq ^ PSWZ(pc))) * 16;
            }
        }
    };

    const uint16_t *wsrc = wbase + lane * 8 + wave * 512;
    auto issue_weights = [&](int st) {   // step st = (chunk st / 3, tap column st % 3): this wave's piece of each of the three tap slices
        const int kc = st / 3, kx = st - kc * 3;
        const int slot = st % 3;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
            glds16s(wsrc + (size_t)(kc * 9 + ky * 3 + kx) * (BCO * 32), s_ring + slot * STEP_BYTES + ky * SLICE_BYTES + wave * 1024);
    };
    auto issue_patch = [&](int kc, bool hf) {
        const bool first = kc < nc0;
        const uint16_t *src = first ? a.in0 : a.in1;
        const unsigned cs = first ? (unsigned)a.C0 : (unsigned)a.C1;
        const int npieces = hf ? (PH0 * PW0 * 4 + 63) / 64 : WPATCH_PIECES;
#pragma unroll
        for (int t = 0; t < PPW; ++t) {
            if (wave + 4 * t >= npieces) break;              // wave-uniform
            int d = pd[t];
            asm volatile("" : "+v"(d));
            const unsigned off = (unsigned)(d >> 5) * cs + (unsigned)((first ? kc : kc - nc0) * 32 + (d & 31));
            glds16s(d >= 0 ? (const void *)(src + off) : zero_page, s_patch + (wave + 4 * t) * 1024);
        }
    };

    f32x4_t acc[TCO][NF];
#pragma unroll
    for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[i][f] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    issue_weights(0);   // S3 >= 6 (two chunks at least)
    issue_weights(1);

    int st = 0;
    bool cur_half = false;
    for (int kc = 0; kc < nchunks; ++kc) {
        const bool half = (kc < nc0) && a.up0;
        const int sh = half ? 1 : 0, row_bytes = (half ? PW0 : PW) * 64;
        if (kc == 0 || half != cur_half) {   // wave-uniform
            build_pd(half);
            build_ct(half);
            cur_half = half;
        }
        // ---- chunk boundary: everyone has left the previous chunk's patch -> refill, drain (also the weights in flight), meet
        if (kc > 0) __builtin_amdgcn_s_barrier();
        issue_patch(kc, half);
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int rowoff[NB / 2];   // patch rows 4*wave + q, q = r + ky = 0..5
#pragma unroll
        for (int q = 0; q < NB / 2; ++q) rowoff[q] = (((4 * wave + q - sh) >> sh) + sh) * row_bytes;
#pragma unroll 1
        for (int kx = 0; kx < 3; ++kx, ++st) {
            if (kx > 0) {
                // this step's three slices have landed (the step after it may be in flight); everyone is done with the slot of step st - 1
                if (st + 1 < S3) wait_vmcnt<3>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
            }
            if (st + 2 < S3) issue_weights(st + 2);
            const char *ws = s_ring + (st % 3) * STEP_BYTES + (fq * BCO + fj) * 16;
            bf16x8_t fb[NB], fa[2][TCO];
            const int c0 = kx == 0 ? ct[0][0] : (kx == 1 ? ct[0][1] : ct[0][2]);
            const int c1 = kx == 0 ? ct[1][0] : (kx == 1 ? ct[1][1] : ct[1][2]);
#pragma unroll
            for (int q = 0; q < NB / 2; ++q) {
                fb[q * 2] = *reinterpret_cast<const bf16x8_t *>(s_patch + rowoff[q] + c0);
                fb[q * 2 + 1] = *reinterpret_cast<const bf16x8_t *>(s_patch + rowoff[q] + c1);
            }
#pragma unroll
            for (int i = 0; i < TCO; ++i) fa[0][i] = *reinterpret_cast<const bf16x8_t *>(ws + i * 256);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                auto mma = [&](int i0, int i1) __attribute__((always_inline)) {
#pragma unroll
                    for (int i = i0; i < i1; ++i)
#pragma unroll
                        for (int f = 0; f < NF; ++f)
                            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ky & 1][i], fb[((f >> 1) + ky) * 2 + (f & 1)], acc[i][f], 0, 0, 0);
                };
                __builtin_amdgcn_sched_barrier(0);
                mma(0, 1);
                __builtin_amdgcn_sched_barrier(0);
                if (ky < 2) {
#pragma unroll
                    for (int i = 0; i < TCO; ++i) fa[(ky + 1) & 1][i] = *reinterpret_cast<const bf16x8_t *>(ws + (ky + 1) * SLICE_BYTES + i * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma(1, TCO);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    stream_epilogue<BCO, TW, SEPI_BF16, NF, false, false, true>(a, acc, co_tile, n, y0, x0, frow, fj, fq, nullptr, nullptr, 0, a.x4 != 0);
}

template <int BCO>
static int launch_wide3(const StreamArgs &a, hipStream_t s) {
    constexpr int smem = 3 * 3 * BCO * 64 + WPATCH_BYTES;   // 75 KiB: two workgroups per CU
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_wide3_kernel<BCO>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    hipLaunchKernelGGL(kern, dim3(a.n_px_tiles * a.n_co_tiles), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_wide3_kernel");
    return V2X_OK;
}

template <int BCO, int EPI>
static int launch_wide(const StreamArgs &a, hipStream_t s) {
    constexpr int smem = RING * BCO * 64 + WPATCH_BYTES + (EPI == SEPI_CHAIN ? chain_lds_bytes<BCO>() : 0);   // 55 KiB (+8.5 chained): two workgroups per CU
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_wide_kernel<BCO, EPI>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    hipLaunchKernelGGL(kern, dim3(a.n_px_tiles * a.n_co_tiles), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_wide_kernel");
    return V2X_OK;
}

// ---- host side -----------------------------------------------------------------------------------
template <int BCO, int TH, int TW, int EPI>
static int launch_stream(const StreamArgs &a, hipStream_t s) {
    constexpr int smem = RING * BCO * 64 + 2 * PATCH_BYTES;  // 80 KiB at BCO=128: two workgroups per CU
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_stream_kernel<BCO, TH, TW, EPI>;
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    hipLaunchKernelGGL(kern, dim3(a.n_px_tiles * a.n_co_tiles), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_stream_kernel");
    return V2X_OK;
}

// split-K pair of launches (small batches: a.ksplit chunk ranges per tile so that few tiles still fill the chip)
template <int BCO, int TH, int TW, int EPI>
static int launch_stream_splitk(const StreamArgs &a, hipStream_t s) {
    constexpr int smem = RING * BCO * 64 + 2 * PATCH_BYTES;
    static v2x_once_per_device attr_once;
    auto kern = &conv3x3_stream_kernel<BCO, TH, TW, SEPI_BF16, true>;   // (the epilogue is the reduce kernel's)
    if (v2x_first_use_on_device(attr_once)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    hipLaunchKernelGGL(kern, dim3(a.n_px_tiles * a.n_co_tiles, a.ksplit), dim3(256), smem, s, a);
    V2X_CHECK_LAUNCH("conv3x3_stream_kernel(split-K)");
    const long long npix = (long long)a.N * a.H * a.W;
    const long long threads = npix * (a.Cout / 4);
    int grid = (int)((threads + 255) / 256 < 4096 ? (threads + 255) / 256 : 4096);
    if (EPI == SEPI_GRU)
        hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3(grid), dim3(256), 0, s, a.ws, a.ksplit, npix, a.w_rows, a.Cout, a.scale, a.shift, a.relu,
                           reinterpret_cast<uint16_t *>(a.out), a.out_cstride, a.out_coff);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(grid), dim3(256), 0, s, a.ws, a.ksplit, npix, a.w_rows, a.Cout, a.scale, a.shift, a.relu,
                           reinterpret_cast<uint16_t *>(a.out), a.out_cstride, a.out_coff);
    V2X_CHECK_LAUNCH("splitk_reduce_kernel");
    return V2X_OK;
}

// the plain-epilogue reduce of a split-K launch, for the stride-2 kernel's split form (conv_stream_s2.hip)
int v2x_launch_splitk_reduce(const float *ws, int ksplit, long long npix, int w_rows, int Cout, const float *scale, const float *shift, int relu,
                             uint16_t *out, int out_cstride, int out_coff, hipStream_t s) {
    const long long threads = npix * (Cout / 4);
    const int grid = (int)((threads + 255) / 256 < 4096 ? (threads + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3(grid), dim3(256), 0, s, ws, ksplit, npix, w_rows, Cout, scale, shift, relu, out, out_cstride, out_coff);
    V2X_CHECK_LAUNCH("splitk_reduce_kernel");
    return V2X_OK;
}

// rows per channel tile the stream kernel uses for (Cout, epilogue); 0 = unsupported
extern "C" int v2x_conv_stream_tile_rows(int Cout, int epilogue) {
    if (epilogue == V2X_EPI_GRU) return (Cout % 32 == 0) ? 96 : 0;
    if (epilogue != V2X_EPI_BF16) return 0;
    if (Cout % 128 == 0) return 128;
    if (Cout % 64 == 0) return 64;
    return 0;
}

// Returns V2X_OK if handled, 1 if the shape is not covered.
int v2x_conv_stream_dispatch(const v2x_conv_desc *d, hipStream_t s) {
    const int rows = v2x_conv_stream_tile_rows(d->Cout, d->epilogue);
    if (rows == 0) return 1;
    const bool t16 = (d->W % 32 != 0);
    if (t16 ? (d->W % 16 != 0 || d->H % 16 != 0) : (d->H % 8 != 0)) return 1;
    StreamArgs a;
    a.in0 = d->in0;
    a.in1 = d->in1;
    a.C0 = d->C0;
    a.C1 = d->C1;
    a.up0 = d->up0;
    a.N = d->N;
    a.H = d->H;
    a.W = d->W;
    a.w = d->weight;
    a.scale = d->scale;
    a.shift = d->shift;
    a.relu = d->relu;
    a.out = d->out;
    a.out_cstride = d->out_cstride;
    a.out_coff = d->out_coff;
    a.Cout = d->Cout;
    a.tiles_x = d->W / (t16 ? 16 : 32);
    a.tiles_y = d->H / (t16 ? 16 : 8);
    a.n_px_tiles = d->N * a.tiles_x * a.tiles_y;
    a.n_co_tiles = d->w_rows / rows;
    a.w2 = d->weight2;
    a.scale2 = d->scale2;
    a.shift2 = d->shift2;
    a.relu2 = d->relu2;
    a.ksplit = d->splitk;
    a.ws = d->splitk_ws;
    a.w_rows = d->w_rows;
    a.xcd_walk = (d->epilogue == V2X_EPI_GRU) ? v2x_tune(V2X_TUNE_GRU_XCD_WALK) : 0;
    // 16-byte output stores (stream_epilogue X4): rows of the output view 16-byte aligned, whole 32-channel pairs of tiles
    a.x4 = (v2x_tune(V2X_TUNE_STORE_X4) != 0 && (d->epilogue == V2X_EPI_BF16 || d->epilogue == V2X_EPI_GRU) && (d->Cout2 > 0 ? d->Cout2 : d->Cout) % 32 == 0 &&
            d->out_cstride % 8 == 0 && d->out_coff % 8 == 0 && (reinterpret_cast<uintptr_t>(d->out) & 15) == 0) ? 1 : 0;
    if (d->w_layout == 4) {   // parity-class form of a decoder `_1` layer (conv_stream_pc.hip): its own kernel, by the packing
        if (t16 || d->H % 16 != 0 || d->up0 != 1 || d->C0 <= 0 || d->C1 <= 0 || (d->Cout % 128 != 0 && d->Cout != 64) || d->epilogue != V2X_EPI_BF16 || d->Cout2 > 0 ||
            d->splitk > 1)
            return 1;
        a.tiles_y = d->H / 16;
        if (d->Cout == 64) {   // conv7_1's shape: two 16 x 32 tiles (a 16 x 64 region) per workgroup
            if (d->W % 64 != 0) return 1;
            a.tiles_x = d->W / 64;
            a.n_co_tiles = 1;
        } else {
            a.n_co_tiles = d->Cout / 128;
        }
        a.n_px_tiles = d->N * a.tiles_x * a.tiles_y;
        return v2x_conv_stream_pc_launch(a, s);
    }
    const bool chain = d->Cout2 > 0;
    if (d->splitk > 1) {
        // small-batch form: the 4-wave kernel with the chunk range divided over blockIdx.y + the reduce kernel.  Plain and GRU epilogues,
        // every split at least one chunk; sums are added in split order (results differ from the unsplit kernels in fp32 summation order)
        const int nchunks = (d->C0 + d->C1) >> 5;
        const int per = (nchunks + d->splitk - 1) / d->splitk;
        if (chain || !d->splitk_ws || d->splitk > nchunks || per * (d->splitk - 1) >= nchunks || (rows != 128 && rows != 96 && rows != 64)) return 1;
        if (d->epilogue == V2X_EPI_GRU) return t16 ? launch_stream_splitk<96, 16, 16, SEPI_GRU>(a, s) : launch_stream_splitk<96, 8, 32, SEPI_GRU>(a, s);
        if (rows == 128) return t16 ? launch_stream_splitk<128, 16, 16, SEPI_BF16>(a, s) : launch_stream_splitk<128, 8, 32, SEPI_BF16>(a, s);
        return t16 ? launch_stream_splitk<64, 16, 16, SEPI_BF16>(a, s) : launch_stream_splitk<64, 8, 32, SEPI_BF16>(a, s);
    }
    if (chain) {  // chained 1x1: 64 -> 64 -> 64 (conv1_2 -> conv3d_1) and 128 -> 128 -> 128 (conv2_2 -> conv3d_2)
        if ((d->Cout != 64 && d->Cout != 128) || d->Cout2 != d->Cout || d->w_rows != d->Cout || d->epilogue != V2X_EPI_BF16 ||
            !d->weight2 || !d->scale2 || !d->shift2)
            return 1;
    }
    // 8-wave ping-pong form: 16x32 tiles, the wide channel tiles.  V2X_STREAM_WAVES=4 forces the 4-wave kernel
    // (A/B runs and the bitwise-equality test; the choice never depends on the batch size).
    // (the chained layer on a handful of maps: its 8-wave form has one channel tile, i.e. N * H * W / 512 workgroups -- 40 for one 5-agent frame at
    // 64 x 64 -- so below 3/4 of a round of the CUs it takes the 4-wave kernel's twice as many, half as large tiles: same K order, same epilogue, the
    // same bits (tests/test_gpu_stream.py::test_stream_chain_conv1x1), which is why this choice may look at the batch)
    const bool few_chain_tiles = chain && (long long)d->N * (d->H / 16) * (d->W / 32) * 4 < 3ll * v2x_num_cus();
    if (!t16 && d->H % 16 == 0 && (rows == 128 || rows == 96) && !few_chain_tiles) {
        if (v2x_tune(V2X_TUNE_STREAM_WAVES) != 4) {
            a.tiles_y = d->H / 16;
            a.n_px_tiles = d->N * a.tiles_x * a.tiles_y;
            // three taps per synchronisation (stream8g) unless V2X_STREAM_G=0 (A/B runs); the chained epilogue keeps the 1-tap form
            const bool grouped = v2x_tune(V2X_TUNE_STREAM_G) != 0 && !chain && a.n_co_tiles <= v2x_num_cus();
            // wave tiling of stream8g: half the channels x 128 pixels per wave for the 128-row layers (-25 % LDS fragment reads: conv5_1
            // 750 -> 710 us, conv6_1 799 -> 751, conv3_2 301 -> 279 per 320 maps, bit-identical); the ConvGRU keeps all channels x 64
            // pixels (with the new tiling 1 685 -> 1 699 us inside the step although +7 % in isolation).  V2X_STREAM_WT=0: old tiling
            // everywhere, =2: new tiling for the GRU too (A/B runs).
            const int wt = v2x_tune(V2X_TUNE_STREAM_WT);
            if (grouped && wt >= 1 && d->epilogue != V2X_EPI_GRU) return launch_stream8g<128, SEPI_BF16, true>(a, s);
            if (grouped && wt >= 2 && d->epilogue == V2X_EPI_GRU) return launch_stream8g<96, SEPI_GRU, true>(a, s);
            if (d->epilogue == V2X_EPI_GRU) return grouped ? launch_stream8g<96, SEPI_GRU>(a, s) : launch_stream8<96, SEPI_GRU>(a, s);
            // (measured and rejected: the chained layer on the three-taps form with the second GEMM's weights read from L2 -- 410 -> 432 us, +5.3 %
            // in the paired A/B: the epilogue's drain and L2 reads cost more than the 12-step layer gains from fewer barriers)
            if (chain) return launch_stream8<128, SEPI_CHAIN>(a, s);
            return grouped ? launch_stream8g<128, SEPI_BF16>(a, s) : launch_stream8<128, SEPI_BF16>(a, s);
        }
    }
    // 64-row layers on maps that tile into 16x32: the wide 4-wave form (V2X_STREAM_WIDE=0 keeps the 256-pixel kernel: A/B)
    if (!t16 && d->H % 16 == 0 && rows == 64 && d->epilogue == V2X_EPI_BF16 && ((d->C0 + d->C1) >> 5) >= 2) {
        // (measured and rejected: the three-taps-per-synchronisation 8-wave form at 64 rows -- 1 172-1 187 us for conv7_1 against
        // 965-979 us for the wide form: 48 MFMAs per step do not cover its load phase)
        if (v2x_tune(V2X_TUNE_STREAM_WIDE) != 0) {
            a.tiles_y = d->H / 16;
            a.n_px_tiles = d->N * a.tiles_x * a.tiles_y;
            // three taps per synchronisation for the plain epilogue and >= 3 chunks (paired A/B: conv7_1, 6 chunks, 507 -> 481 us per 160 maps;
            // a 2-chunk 64 -> 64 layer 276 -> 280 us: the longer step's fill does not amortise); V2X_WIDE3=0 keeps the 1-tap form
            if (!chain && ((d->C0 + d->C1) >> 5) >= 3 && v2x_tune(V2X_TUNE_WIDE3) != 0) return launch_wide3<64>(a, s);
            return chain ? launch_wide<64, SEPI_CHAIN>(a, s) : launch_wide<64, SEPI_BF16>(a, s);
        }
    }
    if (chain) {
        if (d->Cout == 128) return t16 ? launch_stream<128, 16, 16, SEPI_CHAIN>(a, s) : launch_stream<128, 8, 32, SEPI_CHAIN>(a, s);
        return t16 ? launch_stream<64, 16, 16, SEPI_CHAIN>(a, s) : launch_stream<64, 8, 32, SEPI_CHAIN>(a, s);
    }
    if (d->epilogue == V2X_EPI_GRU) return t16 ? launch_stream<96, 16, 16, SEPI_GRU>(a, s) : launch_stream<96, 8, 32, SEPI_GRU>(a, s);
    if (rows == 128) return t16 ? launch_stream<128, 16, 16, SEPI_BF16>(a, s) : launch_stream<128, 8, 32, SEPI_BF16>(a, s);
    return t16 ? launch_stream<64, 16, 16, SEPI_BF16>(a, s) : launch_stream<64, 8, 32, SEPI_BF16>(a, s);
}
