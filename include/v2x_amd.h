/* v2x_amd.h -- C ABI of libv2x_amd.so: the MI355X (gfx950) hot path of the V2X-Sim
 * collaborative-perception baselines.
 *
 * What each entry point replaces.  The reference checkout holds no code: the
 * whole path lives in the un-vendored submodule `coperception`
 * (/root/reference/.gitmodules:1-3; /root/reference/README.md:101 names the
 * det/seg benchmarks, README.md:45 the create_data.py voxelisation).  The
 * reference has no FFI/plugin interface of its own -- its "operator API" is
 * torch.nn.Module.forward -- so every entry below cites the upstream python
 * function (path only; no line numbers exist in the tree) whose arithmetic it
 * takes over.  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions: plain pointers and sizes only (no torch types); all pointers are
 * DEVICE pointers unless marked host; the caller owns every buffer (no internal
 * allocation, no global mutable state); `stream` is a hipStream_t passed as
 * void*; calls are asynchronous on that stream and re-entrant; the return value
 * is 0 or a negative errno-style code, and v2x_last_error() gives the text for
 * the calling thread.  bf16 tensors are passed as uint16_t*.
 *
 * Tensor layouts: activations are NHWC ("BEV pixel major, channels fastest"),
 * which is the reference's own (X, Y, Z) voxel-grid layout with Z as channels.
 */
#ifndef V2X_AMD_H
#define V2X_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2X_AMD_ABI_VERSION 18

#define V2X_OK 0
#define V2X_EINVAL (-22) /* bad argument / unsupported shape */
#define V2X_EIO (-5)     /* HIP launch failure */

typedef void *v2x_stream_t; /* hipStream_t */

/* V2X_AMD_ABI_VERSION of the library; NEGATIVE (-V2X_AMD_ABI_VERSION) when the library contains an instrumented kernel object (a tools/probes/ timing build:
 * phase-removal switches, time stamps -- its results are garbage): a caller that checks the version, as every caller should, then refuses it. */
int v2x_abi_version(void);
const char *v2x_last_error(void);

/* Kernel-selection switches.  The library picks, per layer shape, the kernel form that measured fastest; a few shapes have a second
 * form that computes the same result (the bitwise-equality tests and the paired A/B tool compare them).  `name` is one of
 *   STREAM_WAVES (8 | 4)  STREAM_G  STREAM_WT (0 | 1 | 2)  STORE_X4  STREAM_PERSIST  STREAM_WIDE  WIDE3  HALO_PP  S2_G
 *   VOXELIZE_LDS (0 | 1 | 2)  WARP_LDS  GRU_XCD_WALK  HALO_XCD  WGRAD_TR  BN_PARTIAL_T  WGRAD_REDUCE4  CONV1X1     (0 | 1 unless noted; case-insensitive, an optional "V2X_" prefix is ignored)
 * Each switch is initialised ONCE, at first use, from the environment variable V2X_<NAME> when that is set; afterwards only
 * v2x_tuning_set changes it (process-wide, relaxed atomics: set it before the launches it should affect).  No upstream
 * counterpart (the reference has one implementation per operator).  Unknown name -> V2X_EINVAL. */
int v2x_tuning_set(const char *name, int value);
int v2x_tuning_get(const char *name, int *value);

/* ---------------------------------------------------------------- a1: voxel scatter
 * Replaces coperception/utils/data_util.py::voxelize_occupy (range filter,
 * floor(p / voxel) in fp64, dedupe, dense occupancy) and the densify scatter of
 * coperception/datasets/V2XSimDet.py::__getitem__.
 *
 * pts:   [n_clouds][max_pts][pt_stride] fp32, x,y,z first; cloud i holds n_pts[i] points.
 * extents {xlo,xhi,ylo,yhi,zlo,zhi}, voxel {vx,vy,vz}: HOST fp64 arrays.
 * bits:  [n_clouds][X][Y] uint32, bit z set iff voxel (x,y,z) is occupied (Z <= 32).
 *        The call clears `bits` itself (memset node on `stream`) before scattering. */
int v2x_voxelize_bits(const float *pts, const int32_t *n_pts, int n_clouds, int max_pts, int pt_stride,
                      const double *extents, const double *voxel, const int32_t *dims_xyz,
                      uint32_t *bits, v2x_stream_t stream);

/* Early fusion (the "upperbound" baseline, README.md:101): upstream merges the clouds of all agents into the
 * ego frame at data-preparation time (create_data_det.py, CPU) and voxelizes the union.  Here job j moves source
 * cloud src_cloud[j] by the rigid transform xform[j] (device fp32 [n_jobs][3][4], row-major) and scatters it into
 * target grid dst_grid[j]; several jobs may share a target.  fp32 transform with separately rounded ops in the
 * order ((x*m0 + y*m1) + z*m2) + m3, then the a1 index spec.  bits: [n_grids][X][Y], cleared by the call. */
int v2x_voxelize_fused_bits(const float *pts, const int32_t *n_pts, int n_clouds, int max_pts, int pt_stride,
                            const float *xform, const int32_t *src_cloud, const int32_t *dst_grid, int n_jobs,
                            int n_grids, const double *extents, const double *voxel, const int32_t *dims_xyz,
                            uint32_t *bits, v2x_stream_t stream);

/* bits -> the reference's dense padded_voxel_points layout [n][X][Y][Z] fp32 {0,1}. */
int v2x_bits_to_dense_f32(const uint32_t *bits, int n, int X, int Y, int Z, float *out, v2x_stream_t stream);

/* bits -> network input: NHWC bf16 [n][X][Y][c_pad], channels >= Z are zero. c_pad % 8 == 0. */
int v2x_bits_to_nhwc_bf16(const uint32_t *bits, int n, int X, int Y, int Z, int c_pad, uint16_t *out,
                          v2x_stream_t stream);

/* dense fp32 bevs [n][X][Y][Z] (what the reference Dataset hands the model) -> NHWC bf16 [n][X][Y][c_pad]. */
int v2x_dense_f32_to_nhwc_bf16(const float *bev, int n, int X, int Y, int Z, int c_pad, uint16_t *out,
                               v2x_stream_t stream);

/* bits -> sorted (x,y,z)-lexicographic voxel indices, as voxelize_occupy(return_indices=True).
 * idx: [n][cap][3] int32; counts: [n] int32 (true count even if > cap; only `cap` are written).
 * scratch: [n][X] int32 device workspace. */
int v2x_bits_to_indices(const uint32_t *bits, int n, int X, int Y, int Z, int32_t *idx, int cap, int32_t *counts,
                        int32_t *scratch, v2x_stream_t stream);

/* Densify: the parsed dataset stores each sweep as sparse voxel indices (README.md:66-79 layout;
 * upstream V2XSimDet.__getitem__: curr_voxels[idx[:,0], idx[:,1], idx[:,2]] = 1).  idx: [n][cap][3] int32,
 * counts: [n]; bits as v2x_voxelize_bits (cleared by the call). */
int v2x_indices_to_bits(const int32_t *idx, const int32_t *counts, int n, int cap, int X, int Y, int Z,
                        uint32_t *bits, v2x_stream_t stream);

/* ---------------------------------------------------------------- a2/a4/a6/a7/a8: convolutions
 * One implicit-GEMM MFMA kernel family covers
 *   - coperception/models/det/backbone/Backbone.py::LidarEncoder / LidarDecoder
 *     (3x3 stride 1/2 conv + BN + ReLU; the 1x1x1 "Conv3D"; F.interpolate(x2) +
 *      torch.cat + conv fused through the two-source loader),
 *   - convolutional_rnn.Conv2dGRU one cell step with h0 = 0 (V2VNet.py), gates
 *     fused in the epilogue,
 *   - DetModelBase.py::ClassificationHead / SingleRegressionHead and the seg head,
 *   - When2com.py::PolicyNet4 convs and KmGenerator linears (as 1x1 convs on 1x1 maps).
 */
enum {
    V2X_EPI_BF16 = 0, /* y = acc*scale[c] + shift[c], optional ReLU, bf16 NHWC out          */
    V2X_EPI_F32 = 1,  /* same, fp32 NHWC out (logits)                                        */
    V2X_EPI_GRU = 2,  /* rows are (r,z,n) gate triples; out = (1-z)*n, bf16; see DESIGN.md   */
    V2X_EPI_DET = 3   /* detection heads with the score threshold fused in (halo kernel, chained layer only): instead of the
                       * logits the launch emits the candidates of upstream postprocess.py::apply_nms_det's first step,
                       * softmax(cls)[1] >= det_thr, ready for v2x_det_nms_candidates.  Shape: 3x3 32 -> 64 hidden (cls | reg,
                       * chain order) chained with Cout2 = 64 rows in DET ORDER: packed row 16 t + 4 q + r, q < 3, belongs to
                       * the anchors a0 = 2 q, a1 = 2 q + 1 of the pixel:  t = 0: cls[a0][0], cls[a0][1], cls[a1][0], cls[a1][1];
                       * t = 1: loc[a0][0..3];  t = 2: loc[a0][4], loc[a0][5], loc[a1][0], loc[a1][1];  t = 3: loc[a1][2..5];
                       * rows with q = 3 are zero (6 anchors x (2 + 6) = 48 real rows of 64).  out = keys uint64
                       * [N][det_cap] (~score bits << 32 | anchor index << 12 | slot), out2 = codes fp32 [N][det_cap][6],
                       * det_counts int32 [N] (ZEROED BY THE CALLER; the true count even when > det_cap).  H*W*6 < 2^20. */
};

typedef struct v2x_conv_desc {
    const uint16_t *in0; /* bf16 NHWC [N][H>>up0][W>>up0][C0]                                     */
    const uint16_t *in1; /* bf16 NHWC [N][H][W][C1] or NULL; logical input = cat(up(in0), in1)   */
    int32_t C0, C1;      /* C0 % 8 == 0, C1 % 8 == 0                                              */
    int32_t up0;         /* log2 nearest-neighbour upsample applied to in0 (0 or 1)              */
    int32_t N, H, W;     /* logical input extent                                                 */
    int32_t ksize;       /* 1 or 3                                                               */
    int32_t stride;      /* 1 or 2                                                               */
    int32_t pad;         /* 0 or 1                                                               */
    int32_t Cout;        /* logical output channels (GRU: hidden channels)                       */
    int32_t w_rows;      /* packed weight rows (multiple of v2x_conv_tile_rows)                  */
    int32_t w_kpad;      /* packed K extent, multiple of 64, >= ksize*ksize*(C0+C1)              */
    const uint16_t *weight; /* bf16 [w_rows][w_kpad], k = (ky*ksize+kx)*(C0+C1) + c              */
    const float *scale;  /* fp32 [w_rows]   (GRU: float4 [Cout] = b_ir+b_hr, b_iz+b_hz, b_in, b_hn) */
    const float *shift;  /* fp32 [w_rows]   (GRU: unused)                                        */
    int32_t epilogue;    /* V2X_EPI_*                                                            */
    int32_t relu;
    void *out;           /* NHWC [N][Ho][Wo][out_cstride], written at channel offset out_coff    */
    int32_t out_cstride, out_coff;
    void *out2;          /* optional second NHWC output: channels >= split go to out2[..][c-split] */
    int32_t split;       /* multiple of 4; 0 = no split (out2 ignored)                           */
    int32_t out2_cstride;
    /* --- halo-tile kernel (conv_halo.hip): 3x3 stride-1 layers with <= 96 input channels ------ */
    int32_t w_layout;    /* 4: the same for the STREAMED layers (conv_stream_pc.hip: conv5_1, conv6_1; Cout % 128 == 0, H % 16 == 0,    */
                         /*    W % 32 == 0); w_kpad = 16*C0 + 9*C1                                                                */
                         /* 3: parity-class form of a decoder layer cat(up(in0), in1) -> 3x3 (conv_halo.hip, conv8_1's shape   */
                         /*    C0 = 64, C1 = 32, Cout = 32; H%8==0, W%32==0): see "weight layouts"; w_kpad = 16*C0 + 9*C1     */
                         /* 0: row-major [w_rows][w_kpad] (gather kernel)                          */
                         /* 1: k-slot-major [9*Cin/8][Cout][8] (halo kernel; H%8==0, W%32==0)      */
                         /* 2: streamed slices [co_tile][Cin/32][9][4][rows][8] (conv_stream.hip:    */
                         /*    3x3 stride 1, C0,C1 % 32 == 0, tiles 8x32 / 16x32 / 16x16;           */
                         /*    conv_stream_s2.hip: 3x3 stride 2, one source, H % 8 == 0, W % 64 == 0) */
    int32_t Cout2;       /* > 0: chain a 1x1 conv on the (never stored) Cout-channel result:       */
    const uint16_t *weight2; /* bf16 [ceil16(Cout2)][Cout] row-major; `epilogue`/`split`/out* then  */
    const float *scale2; /*   describe the FINAL output, scale/shift/relu the hidden layer and     */
    const float *shift2; /*   scale2/shift2/relu2 (fp32 [ceil16(Cout2)]) the chained one.          */
    int32_t relu2;       /*   weight rows must be in the chain order ("weight layouts" below).     */
    int32_t in_format;   /* 0: in0 is bf16 NHWC.  1 (w_layout 1, C0 == 32, C1 == 0 only): in0 is the voxelizer's  */
    int32_t in_zbits;    /*    uint32 bit grid [N][H][W]; bit z < in_zbits = channel z, expanded on the fly.      */
    int32_t *det_counts; /* V2X_EPI_DET only (else NULL / 0): candidate counters [N], score threshold, slots per map   */
    float det_thr;
    int32_t det_cap;     /*    <= 4096                                                                              */
    int32_t splitk;      /* > 1 (w_layout 2, stride 1 or 2, no chained layer): small-batch form -- the 32-channel chunks are divided */
    float *splitk_ws;    /*    into `splitk` contiguous ranges computed by different workgroups; fp32 workspace                   */
                         /*    [splitk][N*Ho*Wo][w_rows] (output pixels), caller-owned.  The partial sums are added in range order (deterministic;  */
                         /*    fp32 summation order differs from the unsplit kernels: one bf16 rounding of the output).  Needs    */
                         /*    splitk <= (C0 + C1) / 32 with no empty range: ceil(chunks / splitk) * (splitk - 1) < chunks.        */
    int32_t small_batch; /* 0 (default): the kernel form follows the layer SHAPE only, never N -- a rank that owns fewer items launches the    */
                         /*    same kernels and produces the same bits (R-rank == 1-rank).  1: the CALLER declares a latency launch: forms    */
                         /*    whose choice depends on the tile count may be taken (stride-2 layers: the 1-tap 128-pixel kernel instead of    */
                         /*    the 8-wave three-tap one below 4 tiles per CU; fp32 summation order differs between the two).                 */
} v2x_conv_desc;

/* ---------------------------------------------------------------- weight layouts and their packers (HOST side)
 * A checkpoint holds conv weights as fp32 OIHW [Cout][Cin][k][k] (the ConvGRU: weight_ih [3*hidden][Cin][3][3], gates in
 * (r, z, n) order).  Every kernel reads bf16 (round-to-nearest-even) in one of five layouts, selected by
 * v2x_conv_desc.w_layout.  With Cin' = Cin zero-padded to `cin_pad`, K = k*k*Cin' and the reduction index
 * kk = (ky*k + kx)*Cin' + c  (tap-major, channels fastest, matching NHWC activations):
 *
 *   w_layout 0 (gather kernel, any shape):  [w_rows][w_kpad] row-major; w_rows = Cout rounded up to
 *       v2x_conv_tile_rows(Cout, epilogue), w_kpad = K rounded up to 64, padding = zeros.  scale / shift hold w_rows floats.
 *       GRU: w_rows = hidden/16*48; packed row g*48 + gate*16 + e holds source row gate*hidden + g*16 + e, i.e. the
 *       (r, z, n) rows of 16 hidden channels are adjacent so that one wave owns all three gates of its channels; `scale`
 *       is then float4 [hidden] = (b_ir + b_hr, b_iz + b_hz, b_in, b_hn)  (v2x_pack_gru_bias), `shift` unused.
 *   w_layout 1 (halo kernel, 3x3 stride 1, Cin' % 32 == 0, Cout % 32 == 0):  k-slot-major [K/8][Cout][8]: element
 *       (s, co, j) = W[co][kk = 8*s + j].  w_rows = Cout, w_kpad = K; scale / shift hold Cout floats in natural order.
 *   w_layout 2 (streamed kernels, 3x3, Cin' % 32 == 0, Cout % 64 == 0 or GRU hidden % 32 == 0):  slices
 *       [co_tile][Cin'/32][9 taps][4 slots][rows][8] with rows = v2x_conv_stream_tile_rows(Cout, epilogue): element
 *       (t, ch, tap, slot, r, j) = W[t*rows + r][kk = tap*Cin' + 32*ch + 8*slot + j], followed by 64 B of zeros (the
 *       kernel's zero page).  GRU rows in the (r, z, n)-triple order above (w_rows = 3*hidden); w_kpad = K.
 *   w_layout 3 (parity-class halo kernel; decoder layers F.interpolate(x, 2) + torch.cat + 3x3 of Backbone.py::LidarDecoder, Cin = c_up + C1):
 *       for an output pixel (2Y + py, 2X + px) the three tap rows of the 3x3 read only TWO rows of the half-resolution source -- ky in
 *       G(py, a), a = 0, 1, with G(0,0) = {0}, G(0,1) = {1,2}, G(1,0) = {0,1}, G(1,1) = {2} -- and likewise the columns, so per parity class
 *       (py, px) the upsampled half of the layer is a 2x2-tap convolution on the half-resolution map with the weights
 *       W'[py][px][a][b][co][c] = sum over ky in G(py,a), kx in G(px,b) of W[co][c][ky][kx]  (c < c_up), added in fp32 in (ky, kx) ascending
 *       order and rounded to bf16 ONCE: 4 c_up + 9 C1 instead of 9 (c_up + C1) multiply-adds per output.  Buffer:
 *       [class 2*py+px][tap 2*a+b][c_up/8][Cout][8] followed by the skip half [tap 3*ky+kx][C1/8][Cout][8];  w_rows = Cout,
 *       w_kpad = 16*c_up + 9*C1.  (No upstream counterpart: an exact identity in real arithmetic; the tests hold the kernel to the
 *       unmodified fp32 9-tap layer at the tolerance of the 9-tap kernel.)
 *   w_layout 4 (streamed parity-class kernel; the same identity for Cout % 128 == 0): per 128-row channel tile
 *       [c_up/32 chunks][class tap 2*a+b][class 2*py+px][4 k-slots][128 rows][8] with the pre-summed weights of layout 3, then
 *       [C1/32 chunks][kx][ky][4 k-slots][128 rows][8]; after the last tile 64 B of zeros (the kernel's zero page).  w_rows = Cout,
 *       w_kpad = 16*c_up + 9*C1.
 *   chained layers (Cout2 > 0; layouts 1 and 2): the rows of the FIRST layer are stored in "chain order": packed row
 *       rho = 16*i + 4*q + r holds output channel kappa = 32*(i>>1) + 8*q + 4*(i&1) + r (a lane's accumulators of the first
 *       GEMM are then exactly its B fragment of the second); its scale / shift stay in natural channel order.  The chained
 *       1x1 weights are bf16 [ceil16(Cout2)][Cout] row-major, scale2 / shift2 fp32 [ceil16(Cout2)]  (v2x_pack_chain_1x1).
 *
 * The packers below run on the HOST (no GPU call): query the size, allocate, pack, upload, fill the descriptor. */
typedef struct v2x_pack_spec {
    int32_t Cout;     /* output channels (GRU: hidden channels; the source tensor then has 3*Cout rows) */
    int32_t Cin;      /* input channels of the source tensor                                             */
    int32_t ksize;    /* 1 or 3                                                                          */
    int32_t cin_pad;  /* 0 = Cin, else Cin zero-padded to this many channels (13 -> 32 for the first layer); % 8 == 0 */
    int32_t w_layout; /* 0, 1 or 2 (above)                                                               */
    int32_t epilogue; /* V2X_EPI_*: selects the row tile; V2X_EPI_GRU = (r, z, n) row regrouping           */
    int32_t chain;    /* 1: the layer is followed by a chained 1x1 (Cout2 > 0): rows in chain order       */
    int32_t c_up;     /* w_layout 3 and 4 only: the first c_up input channels are the x2-upsampled source (v2x_conv_desc.C0); else 0 */
    /* ABI 17, v2x_pack_conv_device / _job with transform = 1 only (0, 0 everywhere else): the data-gradient layer of a SLICE of a convolution's input
     * channels -- rows src_row0 .. src_row0 + Cout - 1 of the src_rows rows the full data-gradient layer has (src_rows = the convolution's Cin; the
     * weights are read from its whole tensor W [Cin = this spec's][src_rows][k][k]).  Lets a data gradient be computed as several launches that each
     * fit a kernel (conv8_1 of the decoder: 32 -> 96 channels as 32 -> 64 and 32 -> 32 into one 96-channel map). */
    int32_t src_rows, src_row0;
} v2x_pack_spec;

/* Bytes of the packed bf16 buffer (0 = unsupported spec, see v2x_last_error) and the w_rows / w_kpad to put in the descriptor. */
size_t v2x_pack_conv_size(const v2x_pack_spec *spec, int32_t *w_rows, int32_t *w_kpad);
/* w_oihw: HOST fp32 [rows][Cin][k][k]; dst: HOST buffer of v2x_pack_conv_size bytes. */
int v2x_pack_conv(const v2x_pack_spec *spec, const float *w_oihw, uint16_t *dst);
/* The same packing as v2x_pack_conv ON THE DEVICE, one launch (training re-packs every layer after every optimizer step): w_oihw_dev
 * fp32 DEVICE [rows][Cin][k][k], dst_dev DEVICE buffer of v2x_pack_conv_size bytes, bit-identical to the host packer's output.
 * transform = 1: `spec` describes the DATA-GRADIENT layer of a convolution (spec->Cout = its Cin, spec->Cin = its Cout) and the weights
 * W'[o][c][ky][kx] = W[c][o][2-ky][2-kx] are read from the convolution's own tensor W [spec->Cin][spec->Cout][k][k].  Plain layers only
 * (no V2X_EPI_GRU regrouping, chain = 0). */
int v2x_pack_conv_device(const v2x_pack_spec *spec, const float *w_oihw_dev, int transform, uint16_t *dst_dev, v2x_stream_t stream);
/* Many layers in ONE launch (a training step re-packs ~55 layers after its optimizer step; one launch each was 0.23 ms of a 6-ms step on the
 * launch floor).  v2x_pack_conv_device_job fills ONE job on the HOST for the packing v2x_pack_conv_device would do -- same arguments;
 * block_begin = the sum of the n_blocks of the jobs in front of it -- and returns its block count in *n_blocks.  The caller copies the job array
 * to the device once (jobs hold device pointers: valid while the parameters and destination buffers stay where they are) and calls
 * v2x_pack_conv_device_batch(jobs_dev, n_jobs, total_blocks) after every update.  Bit-identical to the per-layer launches. */
typedef struct v2x_pack_job {
    const float *w;       /* DEVICE fp32 parameter            */
    uint16_t *dst;        /* DEVICE packed bf16 destination   */
    int32_t rows_src, cin, cin_p, taps, K, w_kpad, tile, cout, layout, transform;
    int32_t src_stride, reserved0; /* transform = 1: rows of the source tensor per input channel (= rows_src unless the job packs a row slice)  */
    int64_t groups, data_groups;   /* 16-byte groups of the destination; of them real rows */
    int64_t block_begin;  /* first workgroup of this job in the batched launch */
} v2x_pack_job;
int v2x_pack_conv_device_job(const v2x_pack_spec *spec, const float *w_oihw_dev, int transform, uint16_t *dst_dev, int64_t block_begin,
                             v2x_pack_job *job_host, int64_t *n_blocks);
int v2x_pack_conv_device_batch(const v2x_pack_job *jobs_dev, int32_t n_jobs, int64_t total_blocks, v2x_stream_t stream);
/* Chained 1x1: w2 HOST fp32 [Cout2][Cout] -> dst_w bf16 [ceil16(Cout2)][Cout]; scale2 / shift2 (NULL = ones / zeros)
 * -> fp32 [ceil16(Cout2)], zero beyond Cout2. */
int v2x_pack_chain_1x1(int Cout2, int Cout, const float *w2, const float *scale2, const float *shift2, uint16_t *dst_w,
                       float *dst_scale, float *dst_shift);
/* ConvGRU biases (h0 = 0): bias_ih, bias_hh HOST fp32 [3*hidden] -> dst float4 [hidden] = (b_ir+b_hr, b_iz+b_hz, b_in, b_hn). */
int v2x_pack_gru_bias(int hidden, const float *bias_ih, const float *bias_hh, float *dst);
/* Eval-mode BatchNorm folded behind the fp32 accumulation: scale = gamma / sqrt(var + eps), shift = beta + scale*(conv_bias
 * - mean); gamma == NULL: no BN (scale 1, shift = conv_bias).  Writes n_out >= C floats each, zeros beyond C. */
int v2x_fold_bn(int C, int n_out, const float *conv_bias, const float *gamma, const float *beta, const float *mean,
                const float *var, float eps, float *scale, float *shift);

/* Rows-per-tile the kernel will use for (Cout, epilogue); the weight packer pads w_rows to a multiple. */
int v2x_conv_tile_rows(int Cout, int epilogue);
/* Same for the streamed-weights kernel (w_layout 2); 0 = that kernel does not cover (Cout, epilogue). */
int v2x_conv_stream_tile_rows(int Cout, int epilogue);
int v2x_conv2d(const v2x_conv_desc *desc, v2x_stream_t stream);
/* second(first(x)) for two consecutive HBM-bound layers with the intermediate map kept on chip (conv_halo_pair.hip):
 * replaces Backbone.py::LidarEncoder's conv_pre_1 + bn + relu + conv_pre_2 + bn + relu.  Both descriptors: 3x3, stride 1,
 * pad 1, w_layout 1, C0 = 32 (13 real + zero-weight padding for the first), C1 = 0, Cout = 32, bf16 epilogue;
 * first->in_format = 1 (bit grid, in_zbits <= 16); H % 8 == 0, W % 32 == 0.  first->out is ignored; the result is
 * bit-identical to v2x_conv2d(first) followed by v2x_conv2d(second).
 * SECOND FORM (ABI 16; selected by second->Cout2 > 0 with first->in_format = 0) -- the detection tail, conv_tail.hip: replaces
 * Backbone.py::LidarDecoder's conv8_2 + bn + relu followed by DetModelBase.py's ClassificationHead and SingleRegressionHead.
 * first = conv8_2 (3x3, stride 1, pad 1, w_layout 1, C0 = 32, C1 = 0, Cout = 32, bf16 epilogue, in0 = conv8_1's output, bf16 NHWC [N][H][W][32]);
 * second = the fused heads exactly as v2x_conv2d takes them (w_layout 1, C0 = 32, Cout = 64 hidden rows in chain order, Cout2 = 48, weight2 / scale2 /
 * shift2, relu2 = 0, epilogue V2X_EPI_F32, split = the classification channels (4, 8 or 12), out / out2 16-byte aligned).  H % 8 == 0, W % 32 == 0, N H W < 2^27.  conv8_2's
 * output is never stored (first->out ignored); the logits are bit-identical to v2x_conv2d(first) followed by v2x_conv2d(second). */
int v2x_conv2d_pair(const v2x_conv_desc *first, const v2x_conv_desc *second, v2x_stream_t stream);

/* ---------------------------------------------------------------- f-3: backward of the 3x3 stride-1 convolutions
 * Upstream trains through torch.autograd (cuDNN / MIOpen kernels behind nn.Conv2d.backward, tools/det/train_codet.py).
 *   data gradient:   dX = conv3x3(dY, W') with W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] -- a forward convolution: v2x_conv2d
 *                    on the re-packed weights (v2x_sim_amd/train/hip_conv.py);
 *   weight gradient: v2x_conv3x3_wgrad, an MFMA kernel contracting over pixels (conv_wgrad.hip).
 * x bf16 NHWC [N][H][W][Cin], dy bf16 NHWC [N][H][W][Cout]; H % 8 == 0, W % 32 == 0, Cin % 32 == 0, Cout % 32 == 0.
 * workspace fp32 [n_split][Cout][3][3][Cin]: partial sums, every element written; dW = sum over the first axis (the caller
 * adds them in a fixed order: deterministic).  Cout % 64 == 0: n_split in [1, number of 8x32 pixel tiles]; otherwise (the 32-row
 * form, two workspace slots per block) n_split even, in [2, 2 x tiles].  v2x_conv3x3_wgrad_splits gives the library's choice
 * (0 = unsupported shape).  Stride-2 layers: pass dy with zeros inserted between its pixels (dy_z[2y][2x] = dy[y][x]) and the
 * layer's input x -- the same sum (v2x_sim_amd/train/hip_conv.py). */
int v2x_conv3x3_wgrad_splits(int N, int H, int W, int Cin, int Cout);
int v2x_conv3x3_wgrad(const uint16_t *x, const uint16_t *dy, int N, int H, int W, int Cin, int Cout, float *workspace,
                      int n_split, v2x_stream_t stream);
/* The fixed-order sum of the partials, written in the parameter's own layout: dw_oihw fp32 [Cout][cin_out][3][3] =
 * sum_s workspace[s][co][ky][kx][ci] for ci < cin_out <= Cin (cin_out < Cin: the layer's input was stored zero-padded). */
int v2x_conv3x3_wgrad_reduce(const float *workspace, int n_split, int Cout, int Cin, int cin_out, float *dw_oihw, v2x_stream_t stream);

/* ---------------------------------------------------------------- f-3: batch-statistics BatchNorm + ReLU on bf16 NHWC maps
 * Replaces nn.BatchNorm2d / nn.BatchNorm3d in TRAIN mode followed by F.relu as Backbone.py applies them after every convolution
 * (`F.relu(self.bn1_1(self.conv1_1(x)))`), forward and backward (bn_train.hip).  x [M][C] bf16 with M = N*H*W pixels: the
 * convolution's output as v2x_conv2d wrote it.  C / 8 must divide 256 (C = 8, 16, 32, ..., 2048).
 *   forward:  mean / biased variance over the M pixels (fp32 partials, fp64 finish, fixed order), y = relu((x - mean) * invstd *
 *             gamma + beta) -> bf16; save_mean / save_invstd [C] for the backward; running_mean / running_var (both or neither)
 *             updated as nn.BatchNorm does (momentum, unbiased variance).  relu = 0 leaves the ReLU out.
 *   backward: g = dy * [y > 0] (y recomputed from x), dbeta = sum g, dgamma = sum g * xhat,
 *             dx = gamma * invstd * (g - dbeta / M - xhat * dgamma / M) -> bf16.
 * workspace: v2x_bn_train_workspace_size(M, C) bytes of device memory (per-workgroup partial sums; 0 = unsupported shape).
 * No atomics: results are bit-reproducible run to run. */
long long v2x_bn_train_workspace_size(long long M, int C);
int v2x_bn_train_forward(const uint16_t *x, long long M, int C, const float *gamma, const float *beta, float eps, float momentum,
                         float *running_mean, float *running_var, int relu, uint16_t *y, float *save_mean, float *save_invstd,
                         float *workspace, v2x_stream_t stream);
int v2x_bn_train_backward(const uint16_t *x, const uint16_t *dy, long long M, int C, const float *gamma, const float *beta,
                          const float *save_mean, const float *save_invstd, int relu, uint16_t *dx, float *dgamma, float *dbeta,
                          float *workspace, v2x_stream_t stream);
/* The same, and dx_sum[c] = the sum over the M pixels of dx as stored (bf16-rounded), fp32 [C]: the bias gradient of the convolution whose
 * output x is (autograd's reduction in nn.Conv2d.backward), accumulated by the kernel that writes dx instead of a second pass over it.
 * sum_workspace: v2x_bn_dxsum_workspace_size(M, C) bytes.  Fixed summation order. */
long long v2x_bn_dxsum_workspace_size(long long M, int C);
int v2x_bn_train_backward_dxsum(const uint16_t *x, const uint16_t *dy, long long M, int C, const float *gamma, const float *beta,
                                const float *save_mean, const float *save_invstd, int relu, uint16_t *dx, float *dgamma, float *dbeta,
                                float *dx_sum, float *workspace, float *sum_workspace, v2x_stream_t stream);

/* ---------------------------------------------------------------- a3 (+ the sum of a4/a5): warp + fuse
 * Replaces DetModelBase.py::feature_transformation (affine_grid + grid_sample twice,
 * bilinear, zeros, align_corners=False) together with the reduction that consumes
 * it: torch.mean(torch.stack(neighbours)) in V2VNet.py and the attention-weighted
 * sum of MIMOGeneralDotProductAttention in When2com.py.
 *
 * feat:  bf16 NHWC [A*Bt][H][W][C], item index = agent*Bt + frame (agent-major, as the
 *        reference batches agents).
 * trans: fp32 [Bt][A][A][4][4]; trans[f][ego][nb] is the pose used to bring nb into ego.
 * items: int32 [n_out][2] = (ego agent, frame) of every output map (lets a rank fuse only
 *        the items it owns after the all-gather).
 * coef:  fp32 [n_out][A] weight of source agent j for output m (0 = skip). j == ego is
 *        taken unwarped.
 * mode:  V2X_FUSE_WSUM  out = sum_j coef*warp_j ;  V2X_FUSE_MEAN  out = (sum_{coef!=0} warp_j) / count ;
 *        V2X_FUSE_MAX   out = max_{coef!=0} warp_j  (elementwise; the zero padding of a warped map takes part, as in
 *        upstream's torch.max(torch.stack(...)) of MaxFusion)
 * out:   bf16 NHWC [n_out][H][W][C].   C % 8 == 0. */
enum { V2X_FUSE_WSUM = 0, V2X_FUSE_MEAN = 1, V2X_FUSE_MAX = 2 };
int v2x_warp_fuse(const uint16_t *feat, int A, int Bt, int H, int W, int C, const float *trans,
                  const int32_t *items, int n_out, const float *coef, int mode, uint16_t *out,
                  v2x_stream_t stream);
/* The same launch with its workgroups ordered by FRAME: order int32 [order_len] = the output-map index of (frame slot, ego) in slots of
 * order_stride entries (-1 = none), order_len >= n_out.  The workgroups of XCD x (linear id % 8) compute the slots x, x + 8, ...: the
 * output maps of a frame read the same A source maps, which then cross the fabric once per frame instead of once per ego.  Same arithmetic,
 * identical bits; only the LDS-staged form (H, W multiples of 8, C of 128) uses the table, the direct form ignores it. */
int v2x_warp_fuse_ordered(const uint16_t *feat, int A, int Bt, int H, int W, int C, const float *trans, const int32_t *items, int n_out,
                          const float *coef, int mode, uint16_t *out, const int32_t *order, int order_stride, int order_len,
                          v2x_stream_t stream);

/* ---------------------------------------------------------------- a5: attention handshake
 * Replaces When2com.py::MIMOGeneralDotProductAttention (query projection, key.query
 * scores, softmax over the keys) and the inference-time selection
 * (activated_select: threshold 0.2 / argmax_select: top-1).
 *
 * keys:  fp32 [A*Bt][key_size], querys: fp32 [A*Bt][query_size] (agent-major items)
 * w_lin: fp32 [key_size][query_size], b_lin: fp32 [key_size]   (attention_net.linear)
 * prob:  fp32 [Bt][A(k)][A(q)] softmax scores;  coef: same shape after selection
 * mode:  0 softmax (training / 'softmax'), 1 'activated' (coef = p * (p > thres)), 2 'argmax_test' */
int v2x_attn_handshake(const float *keys, const float *querys, const float *w_lin, const float *b_lin, int A,
                       int Bt, int key_size, int query_size, int mode, float thres, float *prob, float *coef,
                       v2x_stream_t stream);

/* ---------------------------------------------------------------- a8: seg argmax + confusion matrix
 * logits fp32 NHWC [n][H][W][n_cls]; label uint8 [n][H][W]; pred uint8 [n][H][W] (may be NULL);
 * conf int64 [n_cls][n_cls] (rows = label, cols = prediction), accumulated (caller zeroes). */
int v2x_seg_argmax_confusion(const float *logits, const uint8_t *label, int n, int H, int W, int n_cls,
                             uint8_t *pred, long long *conf, v2x_stream_t stream);

/* ---------------------------------------------------------------- f-4 (DiscoNet): per-pixel softmax-weighted fusion
 * Replaces the tail of coperception/models/det/DiscoNet.py::fusion.  scores fp32 [n_items][A][H][W][score_stride] (channel 0
 * = the ReLU'd 1-channel output of PixelWeightedFusionSoftmax for source k), valid fp32 [n_items][A] (0 = source absent),
 * maps bf16 [n_items][A][H][W][C] (source k warped into the ego frame) -> out bf16 [n_items][H][W][C]:
 * w_k = exp(s_k) / sum_j exp(s_j) per pixel, out = sum_k w_k * map_k. */
int v2x_pixel_weighted_fuse(const float *scores, int score_stride, const float *valid, const uint16_t *maps, int n_items,
                            int A, int H, int W, int C, uint16_t *out, v2x_stream_t stream);

/* Per-channel sum of a bf16 [M][C] map -> fp32 [C]: the bias gradient of a convolution (db = dy summed over batch and pixels;
 * upstream: autograd's sum inside nn.Conv2d.backward).  Fixed summation order (bit-reproducible).  C as for the BN entries;
 * workspace: v2x_channel_sum_workspace_size(M, C) bytes (0 = unsupported shape). */
long long v2x_channel_sum_workspace_size(long long M, int C);
int v2x_channel_sum_bf16(const uint16_t *x, long long M, int C, float *out, float *workspace, v2x_stream_t stream);

/* Row f-3 (ABI 17): the gradient of a 1x1 head arriving from the loss -- fp32 [M][C] (C % 4 == 0: 12 class logits, 36 box codes per pixel) -- made ready
 * for the data- and weight-gradient kernels in ONE pass: out = the same values as bf16 [M][Cp], channels C..Cp-1 zero (Cp % 8 == 0, Cp >= C, Cp / 8 divides
 * 256), and sums[c] = the per-channel sum over the M pixels of the fp32 values (the layer's bias gradient; fixed summation order, bit-reproducible).
 * Replaces, in upstream's terms, autograd's pad / cast / sum around nn.Conv2d.backward of ClassificationHead.conv2 and the regression head's last conv
 * (coperception/models/det/base/DetModelBase.py, not in /root/reference): six PyTorch-op launches and four passes over the logit gradients per head.
 * workspace: v2x_cast_pad_chsum_workspace_size(M, Cp) bytes (0 = unsupported shape). */
long long v2x_cast_pad_chsum_workspace_size(long long M, int Cp);
int v2x_cast_pad_chsum_f32(const float *x, long long M, int C, int Cp, uint16_t *out, float *sums, float *workspace, v2x_stream_t stream);

/* Row f-3, the detection loss of a training step, forward and backward (replaces coperception/utils/loss.py's SoftmaxFocalClassificationLoss +
 * WeightedSmoothL1LocalizationLoss as combined by coperception/utils/CoDetModule.py::FaFModule.loss_calculator -- not in /root/reference,
 * README.md:101 names the scripts that call them; restated in v2x_sim_amd/train/loss.py): DEVICE fp32 cls [n][2] logits, labels [n][2] one-hot,
 * loc / targets [n][6] box codes, mask [n] bytes (bool).  cls = sum -alpha_t (1 - p_t)^2 sum_k l_k log softmax(cls)_k with alpha_t = alpha l_1 +
 * (1 - alpha) l_0; loc = sum over masked anchors of smooth-L1 (beta); out4 = {cls / n + loc / n, cls / n, loc / n, n = max(sum l_1, 1)}.
 * Forward = two launches (per-workgroup partials in workspace -- v2x_det_loss_workspace_size(n) bytes -- added in a fixed order: bit-reproducible);
 * backward = one launch writing dcls [n][2] and dloc [n][6] for the incoming gradients of the three outputs (DEVICE scalars, NULL = 0). */
/* Row f-3, the ConvGRU's gate arithmetic in the training graph (convolutional_rnn.Conv2dGRU with hidden = None as V2VNet.py calls it; restated in
 * v2x_sim_amd/train/graph.py::_gru_step): gi fp32 [P][3C][HW] = the input convolution's output (bias_ih included), bias_hh fp32 [3C] ->
 * h fp32 [P][C][HW] = n - z n with r = sigmoid(gi_r + bh_r), z = sigmoid(gi_z + bh_z), n = tanh(gi_n + r bh_n).  Backward: dh -> dgi [P][3C][HW]
 * and dn_r [P][C][HW] = d(pre-activation of n) * r; d bias_hh = (channel sums of dgi's r and z planes, channel sums of dn_r).  HW % 4 == 0. */
int v2x_gru_gates_f32(const float *gi, const float *bias_hh, long long P, int C, int HW, float *h, v2x_stream_t stream);
int v2x_gru_gates_bwd_f32(const float *gi, const float *bias_hh, const float *dh, long long P, int C, int HW, float *dgi, float *dn_r,
                          v2x_stream_t stream);
long long v2x_det_loss_workspace_size(long long n_anchors);
int v2x_det_loss_forward(const float *cls, const float *labels, const float *loc, const float *targets, const uint8_t *mask, long long n_anchors,
                         float alpha, float beta, float *out4, float *workspace, v2x_stream_t stream);
int v2x_det_loss_backward(const float *cls, const float *labels, const float *loc, const float *targets, const uint8_t *mask, long long n_anchors,
                          float alpha, float beta, const float *out4, const float *g_loss, const float *g_cls, const float *g_loc, float *dcls,
                          float *dloc, v2x_stream_t stream);

/* ---------------------------------------------------------------- f-3: the cross-agent warp of the TRAINING graph, forward and data gradient
 * Replaces F.affine_grid + F.grid_sample(mode="bilinear", padding_mode="zeros", align_corners=False) of
 * coperception/models/det/base/IntermediateModelBase.py::feature_transformation (applied twice there: rotation, then translation) and its
 * autograd backward.  in / out fp32 [P][C][H][W] (the fusion stage of the training graph is fp32 NCHW), theta fp32 [P][2][3] (the matrices
 * F.affine_grid takes).  v2x_warp_affine_bwd_f32 is the exact transpose of v2x_warp_affine_f32 with respect to `in` (theta carries no
 * gradient: poses are data), computed as a gather in a fixed order -- bit-reproducible, unlike the atomic scatter of
 * grid_sampler_2d_backward.  Agreement with torch: fp32 rounding of the coordinate arithmetic (tests/test_gpu_train_kernels.py). */
int v2x_warp_affine_f32(const float *in, const float *theta, int P, int C, int H, int W, float *out, v2x_stream_t stream);
int v2x_warp_affine_bwd_f32(const float *dout, const float *theta, int P, int C, int H, int W, float *din, v2x_stream_t stream);

/* f-3 (round 6): V2VNet's message-passing round of the TRAINING graph on bf16 NHWC maps -- no layout or precision change around the fusion stage.
 * Replaces, of coperception/models/det/V2VNet.py::forward (not in /root/reference; README.md:101 names the model; restated in
 * v2x_sim_amd/train/graph.py::v2v_fuse): the per-pair index_select, the two feature_transformation passes (F.affine_grid + F.grid_sample: rotation
 * about the map centre, then translation by (4 T03 / 128, -4 T13 / 128)), the mean over an ego's K neighbours and torch.cat([ego, mean], 1) -- and
 * their autograd backward.  Every item m = 0 .. M-1 has exactly K pairs, consecutive (pair = m K + k).
 *   v2x_v2v_message_bf16: cur, base bf16 [N][H][W][C] (the ego maps; the maps the neighbours send -- the same pointer unless a later round sends the
 *     un-updated maps), trans fp32 [*][4][4], src / tsel int32 [M K] (a pair's row of base / matrix of trans), rows int32 [M] (item m's row of cur)
 *     -> conv_in bf16 [M][H][W][2C] = [cur[rows[m]] | mean_k warp2(base[src[m K + k]])].  The two resampling passes are evaluated without the
 *     intermediate map, with the arithmetic and the order of additions of v2x_warp_affine_f32 applied twice (fp32; one bf16 rounding at the store).
 *   v2x_v2v_message_bwd_bf16: dconv_in bf16 [M][H][W][2C], inv int32 [N][K] (the pairs that read row r of base, in pair order), item_of_row int32 [N]
 *     (the item whose ego map row r is, or -1) -> dbase bf16 [N][H][W][C], the exact transpose (a gather over both passes' candidate pixels in a
 *     fixed order: bit-reproducible); dcur = NULL: the ego half is added into dbase (base == cur), else it is written to dcur [N][H][W][C].
 * The ConvGRU's gates on bf16 NHWC (the arithmetic of v2x_gru_gates_f32): gi bf16 [P][3C] (P = maps x pixels), bias_hh fp32 [3C] -> h bf16 [P][C];
 * backward: dh bf16 [P][C] -> dgi bf16 [P][3C] and sums6c fp32 [6C] = (channel sums of dgi AS STORED: r, z, n = d bias_ih of the input convolution |
 * the r and z sums again, the sums of d(pre-activation of n) * r = d bias_hh), per-workgroup partials in workspace
 * (v2x_gru_gates_nhwc_workspace_size bytes; 0 = unsupported shape) added in a fixed order.  C / 8 must divide 256. */
int v2x_v2v_message_bf16(const uint16_t *cur, const uint16_t *base, const float *trans, const int *src, const int *tsel, const int *rows, int M, int K, int N,
                         int C, int H, int W, uint16_t *conv_in, v2x_stream_t stream);
int v2x_v2v_message_bwd_bf16(const uint16_t *dconv_in, const float *trans, const int *inv, const int *tsel, const int *item_of_row, int M, int K, int N, int C,
                             int H, int W, uint16_t *dbase, uint16_t *dcur, v2x_stream_t stream);
int v2x_gru_gates_nhwc_bf16(const uint16_t *gi, const float *bias_hh, long long P, int C, uint16_t *h, v2x_stream_t stream);
long long v2x_gru_gates_nhwc_workspace_size(long long P, int C);
int v2x_gru_gates_nhwc_bwd_bf16(const uint16_t *gi, const float *bias_hh, const uint16_t *dh, long long P, int C, uint16_t *dgi, float *sums6c,
                                float *workspace, v2x_stream_t stream);

/* f-3 (round 6): the optimizer step.  Adam with torch.optim.Adam's arithmetic (the optimizer upstream's train scripts build, README.md:101:
 *     g = grad (+ weight_decay p);  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps))
 * over up to V2X_ADAM_MAX_TENSORS parameter tensors in ONE launch.  `tensors` is a HOST table of DEVICE fp32 pointers (it travels in the kernel
 * arguments: nothing is uploaded, the launch is capturable); step[i] = the tensor's DEVICE step counter t, already incremented (capturable optimizers),
 * or NULL: t = step_host.  lr_dev = a DEVICE learning rate or NULL: lr.  The hyper-parameters are doubles, as Python holds them: 1 - beta and 1 - beta^t
 * are formed in fp64 ((float)0.999 is 1.3e-5 away from 0.999 in 1 - beta2^t), the per-element arithmetic is fp32.  Tensors with numel 0 are skipped. */
#define V2X_ADAM_MAX_TENSORS 72
typedef struct v2x_adam_tensors {
    float *param[V2X_ADAM_MAX_TENSORS];
    const float *grad[V2X_ADAM_MAX_TENSORS];
    float *exp_avg[V2X_ADAM_MAX_TENSORS];
    float *exp_avg_sq[V2X_ADAM_MAX_TENSORS];
    const float *step[V2X_ADAM_MAX_TENSORS];
    long long numel[V2X_ADAM_MAX_TENSORS];
} v2x_adam_tensors;
int v2x_adam_step_f32(const v2x_adam_tensors *tensors, int n_tensors, const float *lr_dev, double lr, double beta1, double beta2, double eps, double weight_decay,
                      double step_host, v2x_stream_t stream);

/* f-3: the decoder's up + concat of the TRAINING graph and its backward (the inference kernels fold it into their loaders: v2x_conv_desc.up0).
 * Replaces torch.cat((F.interpolate(x, scale_factor=(2, 2)), skip), dim=1) of Backbone.py::LidarDecoder and its autograd backward.
 * lo bf16 [N][H][W][C0] (H, W = the LOW-resolution extent), skip bf16 [N][2H][2W][C1] -> out bf16 [N][2H][2W][C0 + C1] (lo's channels first).
 * Backward: dcat bf16 [N][2H][2W][C0 + C1] -> d_lo [N][H][W][C0] = the 2x2 sums (fp32, row-major order, rounded once), d_skip [N][2H][2W][C1].
 * C0, C1 multiples of 8. */
int v2x_upcat_bf16(const uint16_t *lo, const uint16_t *skip, int N, int H, int W, int C0, int C1, uint16_t *out, v2x_stream_t stream);
int v2x_upcat_bwd_bf16(const uint16_t *dcat, int N, int H, int W, int C0, int C1, uint16_t *d_lo, uint16_t *d_skip, v2x_stream_t stream);
/* f-3: dy bf16 [N][Ho][Wo][C] of a stride-2 layer -> out bf16 [N][2Ho][2Wo][C] with dy at the even positions and zeros elsewhere (the operand of
 * the layer's data and weight gradients as stride-1 computations; replaces torch.zeros + a strided copy).  C % 8 == 0. */
int v2x_zero_insert_bf16(const uint16_t *dy, int N, int Ho, int Wo, int C, uint16_t *out, v2x_stream_t stream);

/* ---------------------------------------------------------------- f-1: detection post-processing
 * Replaces coperception/utils/postprocess.py::apply_nms_det per (agent, frame) map: foreground softmax score, score
 * threshold, 'faf' anchor decode (x, y, w, h, yaw) and greedy NMS on the axis-aligned stand-up boxes, in
 * (score descending, anchor index ascending) order.
 *
 * cls: fp32 [n][M][2] logits;  loc: fp32 [n][M][6] box codes;  anchors: fp32 [M][6] (x, y, w, h, sin, cos), M = X*Y*A.
 * cap: candidate capacity per map, a power of two in [64, 4096].
 * out_boxes fp32 [n][cap][5], out_scores fp32 [n][cap], out_index int32 [n][cap] (anchor index of each detection),
 * out_count int32 [n]: number of detections of map i, or -(number of candidates) when more than `cap` anchors passed the
 * threshold (nothing else is written for that map: the caller retries with a higher threshold or post-processes on the host).
 * key_scratch: uint64 [n][cap], count_scratch: int32 [n] (caller-owned workspace). */
int v2x_det_postprocess(const float *cls, const float *loc, const float *anchors, int n, int M, float score_thr,
                        float nms_thr, int cap, float *out_boxes, float *out_scores, int32_t *out_index,
                        int32_t *out_count, unsigned long long *key_scratch, int32_t *count_scratch,
                        v2x_stream_t stream);

/* Same, but the greedy suppression compares the ROTATED boxes (convex-polygon IoU in fp64; the stand-up overlap is only the
 * cheap reject).  Upstream's apply_nms_det suppresses on stand-up boxes (v2x_det_postprocess); this is the variant SURVEY.md
 * row f-1 lists ("rotated-box NMS"). */
int v2x_det_postprocess_rotated(const float *cls, const float *loc, const float *anchors, int n, int M, float score_thr,
                                float nms_thr, int cap, float *out_boxes, float *out_scores, int32_t *out_index,
                                int32_t *out_count, unsigned long long *key_scratch, int32_t *count_scratch,
                                v2x_stream_t stream);

/* The second half of the two above for candidates that a V2X_EPI_DET launch of v2x_conv2d already selected (the logits never
 * reach memory: apply_nms_det's softmax + threshold run in the heads' epilogue).  keys uint64 [n][cap], codes fp32 [n][cap][6],
 * counts int32 [n] exactly as that launch wrote them; everything else as v2x_det_postprocess (rotated != 0: the rotated-box
 * variant).  Same detections, bit for bit, as v2x_det_postprocess on the logits the heads would have written. */
int v2x_det_nms_candidates(const unsigned long long *keys, const float *codes, const int32_t *counts, const float *anchors, int n,
                           int M, int cap, float nms_thr, int rotated, float *out_boxes, float *out_scores, int32_t *out_index,
                           int32_t *out_count, v2x_stream_t stream);

/* ---------------------------------------------------------------- f-1: the metric (coperception/utils/mean_ap.py::eval_map)
 * Upstream intersects shapely polygons on the host; here the IoU of rotated boxes (x, y, w, h, yaw) is a convex clip in
 * fp64 on the device.
 * v2x_rotated_iou: boxes_a fp32 [na][5], boxes_b fp32 [nb][5] -> iou fp32 [na][nb]. */
int v2x_rotated_iou(const float *boxes_a, int na, const float *boxes_b, int nb, float *iou, v2x_stream_t stream);
/* v2x_match_detections: eval_map's per-image matching (mmdet tpfp_default).  det_boxes fp32 [n_img][det_cap][5] in DESCENDING
 * score order (what v2x_det_postprocess emits), det_count int32 [n_img]; gt_boxes fp32 [n_img][gt_cap][5], gt_count int32
 * [n_img] (gt_cap <= 8192).  Each detection takes the ground truth of highest IoU (lowest index on ties) and is a true positive
 * iff IoU >= iou_thr and that box is still free.  tp int32 [n_img][det_cap] (1 / 0; entries >= det_count untouched),
 * best_iou fp32 [n_img][det_cap] or NULL.  AP itself = a sort + two cumulative sums over (score, tp) of all images: the
 * caller's (v2x_sim_amd/utils/postprocess.py::average_precision). */
int v2x_match_detections(const float *det_boxes, const int32_t *det_count, int det_cap, const float *gt_boxes,
                         const int32_t *gt_count, int gt_cap, int n_img, float iou_thr, int32_t *tp, float *best_iou,
                         v2x_stream_t stream);

/* ---------------------------------------------------------------- d: calibration probes (measurement)
 * No upstream counterpart.  The roofline fractions bench.py prints are graded against the datasheet peaks (8 TB/s, 2.5 PFLOP/s bf16);
 * these two probes measure, on the box the bench runs on, what a pure streaming kernel and a pure MFMA loop sustain, so that fractions
 * and rounds can be compared box-free (bench.py: "calibration").
 * v2x_calib_stream: n_read read streams (src, n_read * units * 16 bytes) and n_write write streams (dst, n_write * units * 16 bytes),
 *   16 bytes per lane, fully coalesced; mixes built: 1:1, 1:3 (the det heads), 2:1, 4:1 (conv1_1), 1:0, 0:1.  nontemporal != 0: nt loads /
 *   stores.  wg_per_cu: persistent workgroups of 256 lanes per CU (<= 0: 8).  Bytes moved = (n_read + n_write) * units * 16.
 * v2x_calib_mfma: one 512-lane workgroup per CU, every wave issues iters * 16 independent v_mfma_f32_16x16x32_bf16 (shape32 != 0: iters * 8
 *   v_mfma_f32_32x32x16_bf16) on register operands; seed != 0: random operand bits (the power-limited rate), 0: constants.  *flops (HOST,
 *   may be NULL) = FLOPs of the launch; clocks (DEVICE uint64[2], may be NULL) = shader cycles and 100-MHz ticks elapsed over one wave's loop
 *   (sustained shader clock in MHz = 100 * clocks[0] / clocks[1]).  scratch: DEVICE, >= 512 floats. */
int v2x_calib_stream(const void *src, void *dst, int64_t units, int n_read, int n_write, int nontemporal, int wg_per_cu,
                     v2x_stream_t stream);
int v2x_calib_mfma(float *scratch, int iters, uint32_t seed, int shape32, uint64_t *clocks, double *flops, v2x_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* V2X_AMD_H */
