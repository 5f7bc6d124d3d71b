/* acx.h -- C ABI of libacx.so: the MI355X-native (gfx950) audio-tagging inference path.
 *
 * The reference (topel/audioset-convnext-inf) has no FFI layer: its boundary for this path is
 * the Python nn.Module surface of `ConvNeXt` (src/audioset_convnext_inf/pytorch/convnext.py).
 * Each entry point below names the reference interface it stands in for (file:line relative to
 * the reference root).  The host-side mirror of that surface lives in
 * audioset-convnext-inf_amd/pytorch/convnext.py and binds these symbols with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / HIP types in any signature
 *     (`stream` is a hipStream_t passed as void*; NULL = the null stream).
 *   - every function returns ACX_OK (0) or a negative acx_status; acx_last_error() gives the
 *     thread-local message of the last failure.
 *   - device pointers are fp32, 16-byte aligned, caller-owned (torch tensors); the context owns
 *     only the repacked weights.  Nothing in the launch path allocates or synchronises, so the
 *     forward can be captured into a hipGraph and `.cpu()` on the outputs synchronises exactly
 *     as it does in the reference.
 *   - activations inside the library are NHWC (channels-last); NCHW appears only in the
 *     frame-embedding output, as in the reference.
 */
#ifndef ACX_H
#define ACX_H

#include <stddef.h>
#include <stdint.h>

#if defined(ACX_BUILD)
#define ACX_API __attribute__((visibility("default")))
#else
#define ACX_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct acx_ctx acx_ctx;

enum acx_status {
    ACX_OK = 0,
    ACX_ERR_ARG = -1,         /* null pointer, bad enum, bad key */
    ACX_ERR_STATE = -2,       /* weights missing / not finalized */
    ACX_ERR_HIP = -3,         /* a HIP runtime call failed */
    ACX_ERR_SHAPE = -4,       /* tensor shape does not match the model, or clip too short */
    ACX_ERR_WORKSPACE = -5,   /* workspace too small or misaligned */
    ACX_ERR_UNSUPPORTED = -6  /* e.g. a device that is not gfx950 */
};

enum acx_mode {
    ACX_MODE_LOGITS = 0,      /* ConvNeXt.forward                  convnext.py:287-331 */
    ACX_MODE_SCENE = 1,       /* ConvNeXt.forward_scene_embeddings convnext.py:333-366 */
    ACX_MODE_FRAME = 2        /* ConvNeXt.forward_frame_embeddings convnext.py:369-402 */
};

/* Arithmetic of the dense contractions (pwconv1/pwconv2, convnext.py:60-62, and the 2x2 downsample convs,
 * convnext.py:230-235).  The reference runs everything in fp32; BASELINE configs[2] asks for "bf16 with fp32
 * LayerNorm": bf16 MFMA operands with fp32 accumulation, while LayerNorm statistics, the residual stream, the
 * depthwise conv, the frontend and the head stay fp32.  The 1e-3 parity bar applies to ACX_PREC_F32 only. */
enum acx_precision {
    ACX_PREC_F32 = 0,         /* v_mfma_f32_32x32x2_f32: fp32 operands on the matrix cores */
    ACX_PREC_BF16 = 1,        /* bf16 operands, fp32 accumulate (NOT within the 1e-3 bar) */
    ACX_PREC_F32_SPLIT = 2,   /* DEFAULT.  fp32 operands carried as fp16 hi + fp16 lo -- v = hi + lo + e with
                               * |e| <= 2^-23 |v| in the worst case (exact for ~40 % of random mantissas; 23 significant bits
                               * where fp32 has 24) -- three fp16 MFMAs per product (the lo*lo term, <= 2^-22 |ab|, is
                               * dropped), fp32 accumulate: fp32-grade results (same parity tests and
                               * tolerances as ACX_PREC_F32, plus tests/test_gpu_stress.py) at 16/3 of the f32-MFMA
                               * rate.  Its kernels launch CU-exclusive workgroups (whole LDS + whole register file of
                               * a CU): work of other streams or processes never shares a CU with them. */
    ACX_PREC_BF16_ACT = 3     /* ACX_PREC_BF16 with the ACTIVATIONS of stages 0-2 (residual stream and depthwise-conv
                               * output: 97 % of the activation bytes) stored in HBM as bf16 as well -- BASELINE configs[2]
                               * read as SURVEY 8d does (107.5 MB of activation traffic per clip).  LayerNorm statistics,
                               * accumulation, GELU, the residual add and the depthwise arithmetic stay fp32; a tensor is
                               * rounded to bf16 (nearest even) when it is written.  Stage 3, frontend and head as in
                               * ACX_PREC_BF16.  The per-layer entry points keep fp32 tensors at the ABI (acx_block converts
                               * at its boundary so that the block runs in this arithmetic). */
};

/* kernel classes for acx_profile_read() */
enum acx_kernel_class {
    ACX_K_FRONTEND = 0, ACX_K_STEM, ACX_K_DWCONV, ACX_K_PW1, ACX_K_PW2, ACX_K_ROWSTATS,
    ACX_K_DOWNSAMPLE, ACX_K_POOLHEAD, ACX_K_TRANSPOSE, ACX_K_MLP_FUSED, ACX_K_MLP_WIDE, ACX_K_COUNT
};

#define ACX_MIN_SAMPLES 7360  /* shortest clip the reference accepts (last 2x2 downsample needs H>=2) */
#define ACX_NUM_CLASSES 527   /* convnext.py:654 */
#define ACX_EMBED_DIM 768     /* convnext.py:656 */

ACX_API const char* acx_last_error(void);
ACX_API int acx_version(void);

/* Lifetime.  One context per process per GPU (reference: one nn.Module placed by
 * `model.to(device)`, demo_convnext.py:44-45, evaluate_convnext_on_audioset.py:45-47). */
ACX_API int acx_create(int hip_device, acx_ctx** out);
ACX_API void acx_destroy(acx_ctx* ctx);

/* Weights enter under the reference's own state_dict keys (the 190-key contract of
 * `load_state_dict` / `safetensors.torch.load_model`, convnext.py:507,
 * evaluate_convnext_on_audioset.py:36-38).  `host_data` is fp32, C-contiguous, HOST memory,
 * borrowed for the duration of the call.  `bn0.num_batches_tracked` (int64) is not needed. */
ACX_API int acx_set_weight(acx_ctx* ctx, const char* state_dict_key, const float* host_data,
                   const int64_t* shape, int ndim);

/* Folds and repacks for the kernels, uploads to the device:
 *   bn0 -> per-mel scale/shift (convnext.py:304-306); melW -> banded form (any matrix: a band may span all 513 bins);
 *   STFT buffers that are window x DFT (torchlibrosa's are hann x DFT) -> the FFT kernel stands in for the two Conv1d
 *   (convnext.py:179-187,298); any other stored buffers -> the dense contraction itself (acx_frontend_info);
 *   dwconv (C,1,7,7) -> [49][C]; LayerNorm affine of each block folded into pwconv1;
 *   gamma folded into pwconv2 (convnext.py:78-83); downsample conv (C',C,2,2) -> [C'][4C] with its
 *   LayerNorm affine folded in.
 * May be called again after weights change. */
ACX_API int acx_finalize(acx_ctx* ctx);

/* Selects enum acx_precision for every later call.  Default: ACX_PREC_F32_SPLIT -- the same default as the Python host
 * (pytorch/convnext.py, ACX_PRECISION overrides it there), so a C caller and a Python caller get the same arithmetic.
 * The reference has no such switch (its torch equivalent is `model.to(torch.bfloat16)` / autocast around the Linear
 * layers).  Changing it drops the finalized state: call acx_finalize again. */
ACX_API int acx_set_precision(acx_ctx* ctx, int precision);

/* Geometry helpers: frames T = L/320+1 (torchlibrosa STFT, hop 320, center) and the spatial size
 * after the stem / each downsample (convnext.py:688-691, 230-235). */
ACX_API int acx_num_frames(int64_t L, int* T);
ACX_API int acx_stage_hw(int64_t L, int stage, int* H, int* W);

ACX_API int acx_workspace_bytes(const acx_ctx* ctx, int B, int64_t L, int mode, size_t* out_bytes);

/* How acx_forward runs a batch of B clips: as *out sub-batches side by side on separate streams (clips are independent in
 * eval mode, convnext.py:219,305 -- BatchNorm uses running statistics; every clip's result is bit-identical however the
 * batch is composed or split).  1 for small batches, while per-kernel profiling is enabled, or with ACX_SPLIT_STREAMS=0;
 * ACX_SPLIT_WAYS=n (1..4) overrides the default of 2.  A forward whose FIRST use of a caller stream happens inside a
 * stream capture (`with torch.cuda.graph(g): model(x)` captures on a private stream) runs un-split -- the side streams
 * cannot be created there -- with the same bits.  No reference counterpart: torch runs one batch on one stream. */
ACX_API int acx_sub_batches(const acx_ctx* ctx, int B, int* out);

/* The hot path.  wav: device (B, L) fp32.
 *   ACX_MODE_LOGITS: out0 = logits (B,527), out1 = probs (B,527)   [dict keys
 *                    "clipwise_logits" / "clipwise_output", convnext.py:329]
 *   ACX_MODE_SCENE : out0 = (B,768), out1 ignored
 *   ACX_MODE_FRAME : out0 = NCHW (B,768,H3,7), out1 ignored */
ACX_API int acx_forward(acx_ctx* ctx, const float* wav, int B, int64_t L, int mode, float* out0,
                float* out1, void* workspace, size_t workspace_bytes, void* stream);

/* ---- multi-GPU: the one collective of the path -----------------------------------------------------------------
 * One process per GPU, full weight replica, clips sharded; logits / probabilities / scene rows are all-gathered over xGMI by
 * RCCL (ncclAllGather), frame embeddings stay sharded.  The reference has no inference-time collective (training DDP only,
 * main.py:641,992-997); the Python host reaches the same collective through torch.distributed (parallel.py).  librccl.so is
 * opened on first use: a single-GPU process never loads it.
 *   rank 0: acx_comm_unique_id(id) -> share the ACX_COMM_ID_BYTES bytes with every rank (file, TCP store, MPI ...)
 *   all   : acx_comm_init(ctx, rank, world, id)         (collective: every rank must call it)
 *   step  : acx_allgather(ctx, send, recv, bytes_per_rank, stream)   recv = world * bytes_per_rank device bytes, rank order;
 *           enqueued on `stream` behind the forward that produced `send` -- no host synchronisation.
 * The communicator belongs to the context (released by acx_destroy). */
#define ACX_COMM_ID_BYTES 128
ACX_API int acx_comm_unique_id(void* id_out);
ACX_API int acx_comm_init(acx_ctx* ctx, int rank, int world, const void* unique_id);
ACX_API int acx_allgather(acx_ctx* ctx, const void* send, void* recv, size_t bytes_per_rank, void* stream);
ACX_API int acx_comm_info(const acx_ctx* ctx, int* rank, int* world);

/* ---- per-kernel entry points (golden-vector tests, rocprofv3 isolation, roofline bench) ---- */

/* K1: Spectrogram + LogmelFilterBank + bn0 (convnext.py:298-306). out (B,T,224). */
ACX_API int acx_logmel_bn0(acx_ctx* ctx, const float* wav, int B, int64_t L, float* out, int apply_bn0,
                   void* stream);
/* K2: stem Conv2d(1,96,4x4,s4,pad(4,0)) + LayerNorm (convnext.py:688-691,227).
 * in (B,T,224) -> out NHWC (B,H0,56,96). */
ACX_API int acx_stem_ln(acx_ctx* ctx, const float* in, int B, int T, float* out, void* stream);
/* K3: Block.dwconv 7x7 depthwise + bias (convnext.py:58-60,76), NHWC in/out, plus the per-pixel
 * LayerNorm statistics (mean, rstd) of the output (convnext.py:78) into stats (B*H*W, 2). */
ACX_API int acx_dwconv7(acx_ctx* ctx, int stage, int block, const float* x, float* y, float* stats, int B,
                int H, int W, void* stream);
/* K3 on bf16 activations -- the storage type of stages 0-2 under ACX_PREC_BF16_ACT (x, y: NHWC bf16, 2 bytes per element;
 * any other precision or stage 3: ACX_ERR_STATE).  Matrix-pipe form: weights rounded to bf16, products exact, fp32
 * accumulation from the bias, one rounding to bf16 at the end (convnext.py:58-60,76). */
ACX_API int acx_dwconv7_bf16(acx_ctx* ctx, int stage, int block, const uint16_t* x, uint16_t* y, int B, int H, int W,
                     void* stream);
/* K3 + K4: whole Block.forward (convnext.py:74-87) on NHWC x, in place: depthwise conv, then LayerNorm + pwconv1 + GELU +
 * pwconv2 + gamma + residual (convnext.py:78-86).  Stages 0-1 (C = 96, 192) run the MLP as ONE fused kernel that keeps
 * the hidden activation in registers; stages 2-3 run two MFMA GEMMs through the hidden scratch.  Set
 * ACX_DISABLE_FUSED_MLP=1 before acx_finalize to force the two-GEMM form everywhere (fp32 precision only).
 * scratch >= acx_block_scratch_bytes, 256-byte aligned. */
ACX_API int acx_block(acx_ctx* ctx, int stage, int block, float* x, int B, int H, int W, void* scratch,
              size_t scratch_bytes, void* stream);
ACX_API int acx_block_scratch_bytes(int stage, int B, int H, int W, size_t* out_bytes);
/* K5: downsample_layers[i], i=1..3: LayerNorm + Conv2d 2x2 s2 (convnext.py:230-235).
 * x NHWC (B,H,W,C_{i-1}) -> out NHWC (B,H/2,W/2,C_i); scratch (B*H*W*C_{i-1}) fp32 receives the
 * normalised copy of x that the 2x2 gather GEMM reads. */
ACX_API int acx_downsample(acx_ctx* ctx, int i, const float* x, float* out, float* scratch, int B, int H,
                   int W, void* stream);
/* K6: pooling + final LayerNorm + head + sigmoid (convnext.py:279-285,321-325).
 * x NHWC (B,H3,7,768). Any of scene/logits/probs may be NULL. */
ACX_API int acx_pool_head(acx_ctx* ctx, const float* x, int B, int H3, float* scene, float* logits,
                  float* probs, void* stream);
/* NHWC -> NCHW (the layout forward_frame_embeddings returns, convnext.py:276-277). */
ACX_API int acx_nhwc_to_nchw(const float* x, float* out, int B, int H, int W, int C, void* stream);

/* Stored clips -> model input, on the device: out[i] = float(double(pcm[i]) / 32767.0), the arithmetic (and therefore the
 * bits) of the reference's host-side `int16_to_float32` (utils/utilities.py:226-227; applied by the AudioSet dataset,
 * utils/data_generator.py:72).  The evaluation sweep copies the int16 clips over PCIe (half the bytes) and widens them here,
 * in stream order in front of acx_forward.  Both buffers on the device, 16-byte aligned; n samples. */
ACX_API int acx_pcm16_to_f32(const int16_t* pcm, float* out, int64_t n, void* stream);

/* Which evaluation of the STFT the frontend uses (round 6).  ACX_FRONTEND_AUTO (default): the FFT kernel when the stored buffers are
 * window x DFT, the dense contraction otherwise (acx_finalize above).  ACX_FRONTEND_DENSE: ALWAYS the dense contraction with the
 * stored `conv_real` / `conv_imag` weights -- the reference's own formulation (two Conv1d, convnext.py:179-187,298) -- 2.1 GFLOP
 * per clip on the f32 matrix cores instead of 0.03.  It is the PARITY MODE for inputs whose spectrum has bins 90 dB and more under
 * the frame peak (pure tones off a bin centre, clean sweeps): such bins hold only the rounding noise of whichever formulation
 * computed them, and the reference's own outputs move by up to 1.4e-3 (frame embeddings) when its two Conv1d are merely
 * accumulated in float64.  Measured against the reference class on nine such probes (tests/test_gpu_frontend_edge.py,
 * profiles/r06_b_frontend_edge.txt): logits / probabilities / scene embeddings within 1e-3 with either frontend; frame embeddings
 * at the reference's own sensitivity with DENSE (<= 1.5e-3), at 2.4-2.7 x it with the FFT (<= 3.6e-3); every other input of the
 * suites is inside 1e-3 with the FFT.  Takes effect at once (re-finalizes a finalized context).  No reference counterpart: the
 * reference has one formulation. */
enum acx_frontend { ACX_FRONTEND_AUTO = 0, ACX_FRONTEND_DENSE = 1 };
ACX_API int acx_set_frontend(acx_ctx* ctx, int mode);

/* How acx_finalize decided to evaluate the frontend: *dense_dft = 0 when the stored STFT buffers are window x DFT (max
 * deviation *stft_deviation <= 2e-6) and the FFT kernel stands in for the two Conv1d, 1 when they are not and the
 * contraction runs as stored (GEMM on the f32 matrix cores) -- the reference applies whatever its state_dict holds
 * (convnext.py:179-187); *mel_taps = stored taps of the banded form of melW (884 for librosa's bank, up to 513 x 224 for a
 * dense matrix, which is applied as it is).  Any pointer may be NULL. */
ACX_API int acx_frontend_info(const acx_ctx* ctx, int* dense_dft, float* stft_deviation, int* mel_taps);

/* Diagnostics.  The tile-shape A/B switches ACX_GEMM_MI, ACX_WIDE_NPB, ACX_WIDE_PERSIST, ACX_GEMM_32X32 and ACX_DW_STREAM are read from the
 * environment once, at the first acx_create; this re-reads them (tests force every tile shape through it and require
 * bit-identical results).  Launches never touch the environment.  No reference counterpart. */
ACX_API int acx_tuning_refresh(void);
/* Test support: make THIS CONTEXT's acx_forward report ACX_ERR_STATE after it has queued sub-batch `sub` (0 .. 3) of a split batch,
 * so that the error path -- joining the forked streams, ending a capture cleanly -- can be exercised; -1 switches it off (the
 * default; acx_finalize also resets it).  Per context (a process-global switch could poison every other user of the
 * library in the process), a call and not an environment variable.  No reference counterpart. */
ACX_API int acx_test_fail_sub(acx_ctx* ctx, int sub);

/* ---- measurement: per-kernel-class device time, HIP events on the launch stream ------------ */
ACX_API int acx_profile_enable(acx_ctx* ctx, int on);
/* Synchronises the recorded events; fills ms[ACX_K_COUNT], launches[ACX_K_COUNT]; resets. */
ACX_API int acx_profile_read(acx_ctx* ctx, double* ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* ACX_H */
