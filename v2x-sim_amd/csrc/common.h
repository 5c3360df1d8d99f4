// Shared helpers for the gfx950 kernels of libv2x_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/v2x_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// thread-local error text behind v2x_last_error()
void v2x_set_error(const char *fmt, ...);

// Kernel-selection switches (include/v2x_amd.h: v2x_tuning_set / v2x_tuning_get).  The DEFAULTS are the measured-fastest forms; the other
// values exist for the bitwise-equality tests and the paired A/B runs.  Each is initialised ONCE from the environment variable V2X_<NAME>
// at first use (api.hip) and changed afterwards only through v2x_tuning_set -- the dispatch never calls getenv.
enum v2x_tune_id {
    V2X_TUNE_STREAM_WAVES,    // 8 (default): 8-wave ping-pong streamed kernels where they apply; 4: the 4-wave kernel everywhere
    V2X_TUNE_STREAM_G,        // 1: three taps per synchronisation (stream8g); 0: the 1-tap 8-wave kernel
    V2X_TUNE_STREAM_WT,       // stream8g wave tiling: 0 all channels x 64 pixels per wave everywhere, 1 (default) half x 128 for the plain layers, 2 also for the ConvGRU
    V2X_TUNE_STORE_X4,        // 1: 16-byte output stores (two channel tiles exchanged with v_permlane16_swap_b32) in the epilogues that have the form; 0: 8-byte stores (same bytes)
    V2X_TUNE_STREAM_PERSIST,  // 1: persistent stream8 grid; 0: one tile per workgroup
    V2X_TUNE_STREAM_WIDE,     // 1: the wide 4-wave form for 64-row layers; 0: the 256-pixel kernel
    V2X_TUNE_WIDE3,           // 1: three taps per synchronisation in the wide form (>= 3 chunks); 0: the 1-tap wide form
    V2X_TUNE_HALO_PP,         // 1: 8-wave ping-pong halo kernel for conv8_1 / conv7_2; 0: the 4-wave kernel
    V2X_TUNE_VOXELIZE_LDS,    // 1: LDS-binned voxeliser when the grid fits; 0: the global-atomic form
    V2X_TUNE_WARP_LDS,        // 2 (default): LDS-staged warp kernel with the rotate set-ups shared through an LDS table; 1: the first LDS-staged form (per-item set-ups); 0: the direct form -- all three bit-identical
    V2X_TUNE_S2_G,            // 1: 8-wave three-tap stride-2 kernel (256-pixel tiles) for the 128-row layers wherever the SHAPE allows (never by batch size; a declared latency launch -- desc->small_batch -- keeps the 1-tap kernel below 4 tiles per CU); 0: the 1-tap kernel
    V2X_TUNE_GRU_XCD_WALK,    // 1: the ConvGRU's persistent grid walks 8 pixel x 4 channel tiles per XCD and round (fabric reads -16 %, same time, same bits); 0: 4 x 8
    V2X_TUNE_HALO_XCD,        // 1: the halo kernels' persistent grids give every XCD a contiguous eighth of the tiles (halo pixels cross the fabric once); 0: round-robin
    V2X_TUNE_WGRAD_TR,        // 1: weight-gradient kernel on LDS-DMA tiles + transpose reads; 0: the first form (VALU transposes)
    V2X_TUNE_BN_PARTIAL_T,    // 1: the training BatchNorm's per-workgroup partial sums stored [kind][channel][workgroup] (the finish kernels read contiguous floats); 0: [workgroup][kind][channel]
    V2X_TUNE_WGRAD_REDUCE4,   // 1: the weight-gradient reduce with 16-byte loads, eight partials in flight (round 6); 0: the scalar form -- bit-identical
    V2X_TUNE_CONV1X1,         // 1: gather-layout 1x1 layers (Cin, Cout <= 128) on the streaming kernel (conv1x1.hip; round 6); 0: the gather kernel -- bit-identical
    V2X_TUNE_COUNT
};
int v2x_tune(int id);

#define V2X_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            v2x_set_error(__VA_ARGS__);   \
            return V2X_EINVAL;            \
        }                                 \
    } while (0)

#define V2X_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            v2x_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return V2X_EIO;                                                      \
        }                                                                        \
    } while (0)

// One-time per-DEVICE setup (hipFuncSetAttribute is a per-device property: a process-global flag would leave the second
// GPU of a process without its > 64 KiB dynamic-LDS opt-in).  hipGetDevice is a thread-local lookup, no driver call.
#define V2X_MAX_DEVICES 32
struct v2x_once_per_device { bool done[V2X_MAX_DEVICES]; };
static inline bool v2x_first_use_on_device(v2x_once_per_device &o) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= V2X_MAX_DEVICES) return true;  // unknown device: just set it again
    if (o.done[dev]) return false;
    o.done[dev] = true;   // benign race: the attribute call is idempotent
    return true;
}

// fp32 -> bf16, round-to-nearest-even (matches torch .to(torch.bfloat16) for finite values)
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// two fp32 -> packed bf16 pair, round-to-nearest-even, in ONE instruction (v_cvt_pk_bf16_f32, new on gfx950).  The
// software form above costs ~12 instructions and two exec-mask round trips (its NaN branch) per value -- measured as
// ~85 instructions per 8-byte output store, more than the MFMA loop of the small-channel layers.
typedef __attribute__((ext_vector_type(2))) __bf16 v2x_bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float v2x_f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const v2x_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, v2x_bf16x2_t));
}

// ReLU on two packed bf16: as 16-bit integers a negative bf16 (sign bit set) is a negative short, so max(x, 0) per half is ONE v_pk_max_i16
// for two values; fmaxf on the fp32 values costs two instructions EACH (it canonicalises its operand first).  relu(round(v)) == round(relu(v)):
// rounding is monotone and -0 maps to +0 either way.
typedef short v2x_s16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t v2x_relu_bf16x2(uint32_t x) {
    const v2x_s16x2_t r = __builtin_elementwise_max(__builtin_bit_cast(v2x_s16x2_t, x), (v2x_s16x2_t){0, 0});
    return __builtin_bit_cast(uint32_t, r);
}
// The same with the floor in a register: floor = 0 -> ReLU, floor = 0x80008000 (two int16 minima) -> identity.  A kernel whose `relu` flag is a
// run-time argument pays ONE instruction per pair either way instead of a max plus a select.
__device__ __forceinline__ uint32_t v2x_relu_bf16x2_floor(uint32_t x, uint32_t floor_bits) {
    const v2x_s16x2_t r = __builtin_elementwise_max(__builtin_bit_cast(v2x_s16x2_t, x), __builtin_bit_cast(v2x_s16x2_t, floor_bits));
    return __builtin_bit_cast(uint32_t, r);
}

// ConvGRU gate arithmetic (a4; upstream calls convgru(x, None): h0 = 0, so h = n + z (0 - n)).  fp32 on the hardware's 1-ulp v_exp_f32 /
// v_rcp_f32: sigmoid(x) = rcp(1 + exp(-x)); tanh(x) = 1 - 2 rcp(1 + exp(2x)), below |x| = 2^-6 the series x (1 - x^2 / 3) (the closed form
// cancels there; the series' own error is x^4 * 2/15 < 8e-9 relative).  Error against libm: a few fp32 ulp for |x| >~ 0.5, growing towards the
// switch point, where 1 - 2 rcp(..) cancels against 1.0: absolute ~1e-7 on a result of ~0.016 = ~1e-5 relative (~100 fp32 ulp) -- still
// 2^-8 of the bf16 rounding that follows (2^-9 relative).  The training graph's gates (gru_train.hip) use libm expf / tanhf: the two agree to that
// 1e-5, not bitwise.  Rounds 1-3 divided in IEEE (v_div_scale / fmas / fixup) and called libm's branchy tanhf: ~64 instructions per hidden value, 9.4 % of
// the ConvGRU kernel's time went to its epilogue (profiles/r04_epilogue_phase.txt); this form is ~22.  ONE definition for every kernel that
// produces GRU output (streamed, gather, split-K reduce): the forms stay bit-consistent with each other.
__device__ __forceinline__ float v2x_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float v2x_tanh(float x) {
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
    const float small = x * (1.0f - x * x * 0.33333334f);
    return __builtin_fabsf(x) < 0.015625f ? small : big;
}
// pre-activations ar, az, an (W_i* x, no bias), b = (b_ir + b_hr, b_iz + b_hz, b_in, b_hn)
__device__ __forceinline__ float v2x_gru_h0(float ar, float az, float an, const float4 &b) {
    const float rg = v2x_sigmoid(ar + b.x);
    const float zg = v2x_sigmoid(az + b.y);
    const float ng = v2x_tanh(an + b.z + rg * b.w);
    return ng + zg * (0.0f - ng);
}

// 16-byte output stores from the 16x16x32 MFMA result layout.  A lane (fj = pixel, fq = k-slot quarter) holds 4 consecutive channels (8 bytes) of
// one pixel per 16-channel tile, and a vector store costs the memory pipeline about the same whatever its width (tools/tile_overhead.py: ~12 us
// per 16 x 32 x 128 tile spent behind dwordx2 stores).  v_permlane16_swap_b32 (gfx950) swaps the odd 16-lane rows of its first operand with the
// even rows of its second: applied to the packed bf16 pairs (x0, y0) of tile i and (x1, y1) of tile i + 1 it leaves the lanes of even fq with 8
// consecutive channels of tile i (theirs, then their neighbour's) and the odd ones with 8 consecutive channels of tile i + 1 (the neighbour's,
// then theirs): ONE dwordx4 store per tile pair, same bytes and values as the two dwordx2 stores.  p = the lane's own 8-byte slot in tile i
// (channel 16 i + 4 fq of its pixel); needs 16-byte aligned rows (channel stride and offset multiples of 8) and whole tile pairs.
__device__ __forceinline__ void v2x_store_pair_x4(uint16_t *p, int fq, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    const auto rx = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(y0, y1, false, false);
    *reinterpret_cast<uint4 *>(p + ((fq & 1) ? 12 : 0)) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
}
// host side: may a layer's bf16 output view take the 16-byte form?
static inline int v2x_x4_ok(const void *out, int out_cstride, int out_coff, int channels) {
    return v2x_tune(V2X_TUNE_STORE_X4) != 0 && channels % 32 == 0 && out_cstride % 8 == 0 && out_coff % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
}

// XCD-aware persistent tile walk for kernels whose neighbouring tiles share input (halo rows / columns): the hardware deals consecutive
// workgroup ids to the 8 XCDs round-robin, so with `tile = blockIdx.x; tile += gridDim.x` the two neighbours of a tile always sit on OTHER XCDs
// and every halo pixel crosses the fabric twice.  Here XCD x (= blockIdx.x % 8) owns a CONTIGUOUS eighth of the tiles and its workgroups stride
// through it: neighbouring tiles (whole maps) share an XCD and its L2.  A pure permutation of the workgroup -> tile map (same balance).
struct v2x_tile_walk { int first, step, end; };
__device__ __forceinline__ v2x_tile_walk v2x_xcd_tile_walk(int n_tiles, int enable) {
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    if (!enable || (G & 7) != 0 || n_tiles < 8) return {b, G, n_tiles};
    const int xcd = b & 7, i = b >> 3, per = G >> 3;
    const int q = n_tiles >> 3, r = n_tiles & 7;
    const int lo = xcd * q + (xcd < r ? xcd : r);
    return {lo + i, per, lo + q + (xcd < r ? 1 : 0)};
}
