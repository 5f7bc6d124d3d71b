// K1 -- waveform -> log-mel (+ bn0) in one kernel.
//
// Stands in for torchlibrosa Spectrogram + LogmelFilterBank as the reference constructs and calls
// them (convnext.py:179-200, :298-299) followed by bn0 (convnext.py:304-306):
//   reflect-pad 512|512, frame t = padded[320 t, 320 t + 1024), hann window, |DFT|^2 (513 bins),
//   mel = P . melW (513x224, banded), 10 log10(max(mel, 1e-10)), per-mel affine of eval BatchNorm.
// The reference evaluates the DFT as two dense Conv1d (2.1 GFLOP/clip); here each wave64 runs a
// 1024-point real FFT in LDS (radix-8 Stockham, fft_core.h): 64 lanes x 8 complex points.
// HBM traffic: wav in (L*4 B/clip, frames overlap 3.2x but hit L2), (T,224) fp32 out.
#include "acx_internal.h"
#include "fft_core.h"

namespace acx {

constexpr int kFrontWaves = 4;


// Mel weights in the LDS.  Fast form (round 4): bin m = lane + 64 i of group i has at most kMelTaps[i] taps -- true of librosa's
// 224-bin bank at 32 kHz (1 / 3 / 8 / 14) -- and its weights sit zero-padded at [group base + 64 q + lane]: four unrolled loops
// of fixed length whose 52 LDS reads are all in flight together, instead of four run-time loops of 26 iterations in all, each a
// dependent LDS round trip (a fifth of a frame's time at two waves per SIMD).  A padded tap multiplies a finite spectrum value
// (or, past bin 512, finite second-pass data of the same frame) by zero: same sums, same bits.  Any other bank takes the general
// loops over the banded table (kept in the LDS when it fits).
constexpr int kMelTaps[4] = {1, 3, 8, 14};
constexpr int kMelBase[5] = {0, 64, 256, 768, 1664};
constexpr int kMelLds = 1664;

struct __attribute__((packed, aligned(4))) FrontF2 { float a, b; };

struct FrontLds {
    float melw[kMelLds];
    cf tw[1024];
    cf buf[kFrontWaves][2][kFftBufSlots];      // buf[w][1] doubles as the power spectrum P (513 floats) once pass 3 has
};                                             // read it: 49 KB per workgroup = 3 workgroups per CU

__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wav, long long L, int T,
                                                     long long nframes, const float* __restrict__ hann,
                                                     const float* __restrict__ twiddle,
                                                     const int* __restrict__ mel_start,
                                                     const int* __restrict__ mel_len,
                                                     const int* __restrict__ mel_off,
                                                     const float* __restrict__ mel_w, int mel_w_len,
                                                     const float* __restrict__ bn_scale,
                                                     const float* __restrict__ bn_shift, float* __restrict__ out) {
    __shared__ FrontLds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    for (int i = tid; i < 1024; i += 256) lds.tw[i] = cf_make(twiddle[2 * i], twiddle[2 * i + 1]);
    const bool mel_in_lds = mel_w_len <= kMelLds;      // else the filter loop reads the weights from global memory

    // this lane's mel bins (224 = 3.5 x 64) and their band descriptors
    int mstart[4], mlen[4], moff[4];
    float msc[4], msh[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = lane + 64 * i;
        bool ok = m < kMels;
        mstart[i] = ok ? mel_start[m] : 0;
        mlen[i] = ok ? mel_len[m] : 0;
        moff[i] = ok ? mel_off[m] : 0;
        msc[i] = ok ? bn_scale[m] : 0.f;
        msh[i] = ok ? bn_shift[m] : 0.f;
    }
    // every wave holds all 224 bins (lane + 64 i): the choice of the filter loop is the same in every wave
    // (the unrolled loops read kMelTaps[i] power values from mstart[i] on for EVERY bin and rely on zero weights for the padded
    // taps: a padded tap past the 513 power bins the frame wrote would multiply uninitialised LDS -- NaN bits times 0 is NaN --
    // so a bank with such a bin takes the general loops)
    const bool mel_fast = __all(mlen[0] <= kMelTaps[0] && mlen[1] <= kMelTaps[1] && mlen[2] <= kMelTaps[2] && mlen[3] <= kMelTaps[3] &&
                                mstart[0] + kMelTaps[0] <= kBins && mstart[1] + kMelTaps[1] <= kBins &&
                                mstart[2] + kMelTaps[2] <= kBins && mstart[3] + kMelTaps[3] <= kBins);
    if (mel_fast) {
        for (int e = tid; e < kMelLds; e += 256) {
            const int i = e < kMelBase[1] ? 0 : (e < kMelBase[2] ? 1 : (e < kMelBase[3] ? 2 : 3));
            const int rel = e - kMelBase[i], q = rel >> 6, m = (rel & 63) + 64 * i;
            float wv = 0.f;
            if (m < kMels && q < mel_len[m]) wv = mel_w[mel_off[m] + q];
            lds.melw[e] = wv;
        }
    } else if (mel_in_lds) {
        for (int i = tid; i < mel_w_len; i += 256) lds.melw[i] = mel_w[i];
    }
    float2 hw[8];   // hann for this lane's 16 samples (same for every frame)
#pragma unroll
    for (int r = 0; r < 8; ++r) hw[r] = *reinterpret_cast<const float2*>(hann + 2 * (lane + 64 * r));
    __syncthreads();
    // The twiddles of the two exchange passes and of the real split depend on the lane only: 22 complex registers instead
    // of 22 LDS reads per frame whose strides (128 r bytes in pass Ns = 8) were this kernel's bank conflicts.
    cf tw8[7], tw64[7], twk[8];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
        tw8[r - 1] = lds.tw[2 * ((lane % 8) * 8) * r];
        tw64[r - 1] = lds.tw[2 * lane * r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) twk[r] = lds.tw[lane + 64 * r];
    const cf tw512 = lds.tw[512];
    // A wave works on its own frame in its own two buffers; the LDS serves one wave's accesses in issue order, so between
    // a wave's writes and its reads of other lanes' data nothing but the compiler has to be held back: no workgroup
    // barrier in the frame loop (the four waves used to move in lockstep through four of them per frame).
#define ACX_WAVE_LDS_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

    const long long per_iter = (long long)gridDim.x * kFrontWaves;
    const long long iters = (nframes + per_iter - 1) / per_iter;
    for (long long it = 0; it < iters; ++it) {
        const long long f = it * per_iter + (long long)blockIdx.x * kFrontWaves + wave;
        const bool valid = f < nframes;
        const long long b = valid ? f / T : 0;
        const int t = valid ? (int)(f - b * T) : 0;
        const float* x = wav + b * L;
        const long long p0 = 320LL * t;

        cf v[8];
        const bool interior = (p0 >= 512) && (p0 + 512 <= L);
        if (interior) {
            const float* xs = x + (p0 - 512);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int n = 2 * (lane + 64 * r);
                const FrontF2 xv = *reinterpret_cast<const FrontF2*>(xs + n);     // one 8-byte load (4-byte aligned: clips of odd length)
                v[r] = cf_make(xv.a * hw[r].x, xv.b * hw[r].y);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                int n = 2 * (lane + 64 * r);
                float a = x[reflect_index(p0 + n, L)];
                float c = x[reflect_index(p0 + n + 1, L)];
                v[r] = cf_make(a * hw[r].x, c * hw[r].y);
            }
        }
        // pass Ns=1 straight from registers, then two LDS-exchange passes
        cf* bufA = lds.buf[wave][0];
        cf* bufB = lds.buf[wave][1];
        ACX_WAVE_LDS_SYNC();        // the previous frame's reads of bufA / P are done
        int dst = fft512_pass(v, lane, 1, lds.tw);
#pragma unroll
        for (int r = 0; r < 8; ++r) bufA[fft_pad(dst + r)] = v[r];
        ACX_WAVE_LDS_SYNC();
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = bufA[fft_pad(lane + 64 * r)];
        dst = fft512_pass_tw(v, lane, 8, tw8);
#pragma unroll
        for (int r = 0; r < 8; ++r) bufB[fft_pad(dst + r * 8)] = v[r];
        ACX_WAVE_LDS_SYNC();
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = bufB[fft_pad(lane + 64 * r)];
        dst = fft512_pass_tw(v, lane, 64, tw64);
        ACX_WAVE_LDS_SYNC();        // every lane has read bufA's first-pass data before it is rewritten
#pragma unroll
        for (int r = 0; r < 8; ++r) bufA[fft_pad(dst + r * 64)] = v[r];
        ACX_WAVE_LDS_SYNC();
        // power spectrum, 513 bins
        float* P = reinterpret_cast<float*>(bufB);       // bufB was last read before the wait above
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int k = lane + 64 * r;
            cf X = rfft1024_bin_tw(bufA, k, twk[r]);
            P[k] = X.x * X.x + X.y * X.y;
        }
        if (lane == 0) {
            cf X = rfft1024_bin_tw(bufA, 512, tw512);
            P[512] = X.x * X.x + X.y * X.y;
        }
        ACX_WAVE_LDS_SYNC();
        // banded mel filter + dB + bn0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = 0.f;
            const float* p = P + mstart[i];
            if (mel_fast) {
                const float* w = lds.melw + kMelBase[i] + lane;
#pragma unroll
                for (int q = 0; q < kMelTaps[i]; ++q) acc = fmaf(p[q], w[64 * q], acc);
            } else if (mel_in_lds) {
                const float* w = lds.melw + moff[i];
                for (int q = 0; q < mlen[i]; ++q) acc = fmaf(p[q], w[q], acc);
            } else {
                const float* w = mel_w + moff[i];
                for (int q = 0; q < mlen[i]; ++q) acc = fmaf(p[q], w[q], acc);
            }
            float db = 10.0f * log10f(fmaxf(acc, 1e-10f));
            int m = lane + 64 * i;
            if (valid && m < kMels) out[f * kMels + m] = fmaf(db, msc[i], msh[i]);
        }
        // bufA / P are rewritten only after the next iteration's first wait
    }
#undef ACX_WAVE_LDS_SYNC
}


// ---- dense-DFT fallback: STFT buffers that are NOT window x DFT (acx_finalize, api.hip) -----------------------------------
// The reference evaluates spectrogram_extractor.stft.conv_real / conv_imag as two Conv1d over the reflect-padded waveform
// whatever they hold (convnext.py:179-187, :298); when the FFT cannot stand in for them the same contraction runs as
//   frames [B T, 1024] (reflect-padded, NO window: it is inside the stored weights) . [conv_real; conv_imag]^T -> [B T, 2 x 513]
// on the f32-input matrix cores (gemm.hip, the native fp32 GEMM: in every precision mode the frontend stays fp32), then
// |.|^2, the banded mel filter, dB and bn0 exactly as in logmel_kernel.
__global__ __launch_bounds__(256) void frames_kernel(const float* __restrict__ wav, long long L, int T, long long nframes,
                                                     float* __restrict__ frames) {
    for (long long f = blockIdx.x; f < nframes; f += gridDim.x) {
        const long long b = f / T;
        const int t = (int)(f - b * T);
        const float* x = wav + b * L;
        const long long p0 = 320LL * t;
        float4 v;
        const int n = 4 * threadIdx.x;
        v.x = x[reflect_index(p0 + n, L)]; v.y = x[reflect_index(p0 + n + 1, L)];
        v.z = x[reflect_index(p0 + n + 2, L)]; v.w = x[reflect_index(p0 + n + 3, L)];
        reinterpret_cast<float4*>(frames + f * kNFFT)[threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void spec_to_logmel_kernel(const float* __restrict__ spec /*[nframes][kDenseN]*/, long long nframes,
                                                             const int* __restrict__ mel_start, const int* __restrict__ mel_len,
                                                             const int* __restrict__ mel_off, const float* __restrict__ mel_w,
                                                             const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
                                                             float* __restrict__ out) {
    __shared__ float P[kFrontWaves][kBins + 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long f = (long long)blockIdx.x * kFrontWaves + wave; f < nframes; f += (long long)gridDim.x * kFrontWaves) {
        const float* sp = spec + f * kDenseN;
        for (int k = lane; k < kBins; k += 64) {
            const float re = sp[k], im = sp[kBins + k];
            P[wave][k] = re * re + im * im;                  // torchlibrosa: real ** 2 + imag ** 2 (power = 2.0)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (int m = lane; m < kMels; m += 64) {
            float acc = 0.f;
            const float* p = P[wave] + mel_start[m];
            const float* w = mel_w + mel_off[m];
            const int n = mel_len[m];
            for (int q = 0; q < n; ++q) acc = fmaf(p[q], w[q], acc);
            const float db = 10.0f * log10f(fmaxf(acc, 1e-10f));
            out[f * kMels + m] = fmaf(db, bn_scale[m], bn_shift[m]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // P is rewritten by the next frame
    }
}

static int launch_logmel_dense(acx_ctx* c, const float* wav, int B, int64_t L, int T, float* out, bool bn, hipStream_t s,
                               float* frames, float* spec) {
    const long long nframes = (long long)B * T;
    // per-kernel entry point (acx_logmel_bn0; the forward passes its own workspace): frames + spectrum of THIS call, allocated and
    // released in stream order on `s` -- calls on different streams or threads never share scratch (one grown-on-demand
    // buffer per context was shared by every call in flight)
    void* call_scratch = nullptr;
    if (!frames || !spec) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (s != nullptr && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
            ACX_FAIL(ACX_ERR_STATE, "acx_logmel_bn0 (dense-DFT frontend) cannot run inside a stream capture: use acx_forward");
        const size_t need = (size_t)nframes * (kNFFT + kDenseN) * 4;
        ACX_HIP(hipMallocAsync(&call_scratch, need, s));
        frames = reinterpret_cast<float*>(call_scratch);
        spec = frames + (size_t)nframes * kNFFT;
    }
    struct ScratchFree {        // every exit below (error returns included) hands the bytes back behind the launches queued so far
        void* p; hipStream_t s;
        ~ScratchFree() { if (p) (void)hipFreeAsync(p, s); }
    } scratch_free{call_scratch, s};
    ProfScope ps(c, ACX_K_FRONTEND, s);
    long long blocks = nframes < 4096 ? nframes : 4096;
    launch_kernel(&frames_kernel, dim3((unsigned)blocks), dim3(256), 0, s, wav, L, T, nframes, frames);
    ACX_HIP(hipGetLastError());
    GemmArgs g{};
    g.A = frames; g.Wt = c->d_stft_w; g.bias = c->d_stft_zero; g.out = spec; g.M = nframes; g.N = kDenseN; g.K = kNFFT;
    g.epi = EPI_BIAS; g.cls = -1;       // (inside this ProfScope: not a class of its own)
    ACX_TRY(launch_gemm(nullptr, g, s));
    blocks = (nframes + kFrontWaves - 1) / kFrontWaves;
    if (blocks > 2048) blocks = 2048;
    launch_kernel(&spec_to_logmel_kernel, dim3((unsigned)blocks), dim3(256), 0, s, spec, nframes, c->d_mel_start, c->d_mel_len, c->d_mel_off, c->d_mel_w,
                                                                        bn ? c->d_bn_scale : c->d_bn_one, bn ? c->d_bn_shift : c->d_bn_zero, out);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

int launch_logmel(acx_ctx* c, const float* wav, int B, int64_t L, int T, float* out, bool bn, hipStream_t s,
                  float* dense_frames, float* dense_spec) {
    if (c->dense_stft) return launch_logmel_dense(c, wav, B, L, T, out, bn, s, dense_frames, dense_spec);
    long long nframes = (long long)B * T;
    long long blocks = (nframes + kFrontWaves - 1) / kFrontWaves;
    // 6 workgroups per CU (3 resident): a workgroup's setup -- twiddle and mel tables into the LDS, lane twiddles into registers --
    // is paid once per ~10 frames of a wave (with 4 096 workgroups it was once per 4: 184 -> 167 us at B = 64)
    if (blocks > 1536) blocks = 1536;
    ProfScope ps(c, ACX_K_FRONTEND, s);
    launch_kernel(&logmel_kernel, dim3((unsigned)blocks), dim3(256), 0, s, wav, L, T, nframes, c->d_hann, c->d_twiddle,
                                                                c->d_mel_start, c->d_mel_len, c->d_mel_off,
                                                                c->d_mel_w, c->mel_w_len, bn ? c->d_bn_scale : c->d_bn_one,
                                                                bn ? c->d_bn_shift : c->d_bn_zero, out);
    ACX_HIP(hipGetLastError());
    return ACX_OK;
}

}  // namespace acx
