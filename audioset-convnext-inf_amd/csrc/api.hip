// libacx C ABI: context lifetime, weight intake / folding / repacking, workspace planning and the
// forward orchestration.  See include/acx.h for the contract and the reference interfaces each
// entry point stands in for.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "acx_internal.h"

namespace acx {

static thread_local char g_err[512] = "";
thread_local int tls_inflight_ways = 1;

Tuning& tuning() {
    static Tuning t;
    return t;
}
void tuning_reload() {
    auto digit = [](const char* name, const char* allowed, int none) {
        const char* e = std::getenv(name);
        if (!e || !e[0] || e[1] || !std::strchr(allowed, e[0])) return none;
        return e[0] - '0';
    };
    Tuning& t = tuning();
    t.gemm_mi.store(digit("ACX_GEMM_MI", "124", 0), std::memory_order_relaxed);
    t.wide_npb.store(digit("ACX_WIDE_NPB", "12", 0), std::memory_order_relaxed);
    t.wide_pers.store(digit("ACX_WIDE_PERSIST", "01", -1) < 0 ? 0 : (digit("ACX_WIDE_PERSIST", "01", 0) == 1 ? 1 : 2), std::memory_order_relaxed);
    t.gemm_32x32.store(digit("ACX_GEMM_32X32", "1", 0), std::memory_order_relaxed);
    t.dw_stream.store(digit("ACX_DW_STREAM", "01", -1), std::memory_order_relaxed);
    t.dw_mfma.store(digit("ACX_DW_MFMA", "01", -1), std::memory_order_relaxed);
    t.dwm_waves.store(digit("ACX_DWM_WAVES", "23456789", 0), std::memory_order_relaxed);
}

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local ProfScope* g_prof_scope = nullptr;

ProfScope::ProfScope(acx_ctx* c, int k, hipStream_t st) : ctx(c), cls(k), s(st), prev(g_prof_scope) {
    if (!ctx || !ctx->prof.on || cls < 0) ctx = nullptr;      // (an inner scope without a class keeps the outer one's)
    if (ctx) g_prof_scope = this;
}
ProfScope::~ProfScope() {
    if (ctx) g_prof_scope = prev;
}

void prof_next_events(hipEvent_t* a, hipEvent_t* b) {
    ProfScope* sc = g_prof_scope;
    if (!sc) return;
    acx_ctx* ctx = sc->ctx;
    auto get = [&]() {
        hipEvent_t e;
        if (!ctx->prof.pool.empty()) { e = ctx->prof.pool.back(); ctx->prof.pool.pop_back(); }
        else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        return e;
    };
    *a = get(); *b = get();
    if (!*a || !*b) { if (*a) ctx->prof.pool.push_back(*a); if (*b) ctx->prof.pool.push_back(*b); *a = *b = nullptr; return; }
    ctx->prof.recs.push_back({sc->cls, *a, *b});
}

struct KeySpec { std::string key; std::vector<int64_t> shape; };

static std::vector<KeySpec> required_keys() {
    std::vector<KeySpec> v;
    v.push_back({"spectrogram_extractor.stft.conv_real.weight", {kBins, 1, kNFFT}});
    v.push_back({"spectrogram_extractor.stft.conv_imag.weight", {kBins, 1, kNFFT}});
    v.push_back({"logmel_extractor.melW", {kBins, kMels}});
    for (const char* k : {"bn0.weight", "bn0.bias", "bn0.running_mean", "bn0.running_var"}) v.push_back({k, {kMels}});
    v.push_back({"downsample_layers.0.0.weight", {kDims[0], 1, 4, 4}});
    v.push_back({"downsample_layers.0.0.bias", {kDims[0]}});
    v.push_back({"downsample_layers.0.1.weight", {kDims[0]}});
    v.push_back({"downsample_layers.0.1.bias", {kDims[0]}});
    char buf[96];
    for (int i = 1; i < 4; ++i) {
        snprintf(buf, sizeof buf, "downsample_layers.%d.0.weight", i); v.push_back({buf, {kDims[i - 1]}});
        snprintf(buf, sizeof buf, "downsample_layers.%d.0.bias", i); v.push_back({buf, {kDims[i - 1]}});
        snprintf(buf, sizeof buf, "downsample_layers.%d.1.weight", i); v.push_back({buf, {kDims[i], kDims[i - 1], 2, 2}});
        snprintf(buf, sizeof buf, "downsample_layers.%d.1.bias", i); v.push_back({buf, {kDims[i]}});
    }
    for (int s = 0; s < 4; ++s) {
        const int64_t C = kDims[s];
        for (int j = 0; j < kDepths[s]; ++j) {
            auto key = [&](const char* leaf) { snprintf(buf, sizeof buf, "stages.%d.%d.%s", s, j, leaf); return std::string(buf); };
            v.push_back({key("gamma"), {C}});
            v.push_back({key("dwconv.weight"), {C, 1, 7, 7}});
            v.push_back({key("dwconv.bias"), {C}});
            v.push_back({key("norm.weight"), {C}});
            v.push_back({key("norm.bias"), {C}});
            v.push_back({key("pwconv1.weight"), {4 * C, C}});
            v.push_back({key("pwconv1.bias"), {4 * C}});
            v.push_back({key("pwconv2.weight"), {C, 4 * C}});
            v.push_back({key("pwconv2.bias"), {C}});
        }
    }
    v.push_back({"norm.weight", {kDims[3]}});
    v.push_back({"norm.bias", {kDims[3]}});
    v.push_back({"head_audioset.weight", {kClasses, kDims[3]}});
    v.push_back({"head_audioset.bias", {kClasses}});
    return v;
}

static const std::vector<KeySpec>& key_table() {
    static const std::vector<KeySpec> t = required_keys();
    return t;
}

template <typename T>
static int upload(acx_ctx* c, const std::vector<T>& h, T** out) {
    void* d = nullptr;
    ACX_HIP(hipMalloc(&d, h.size() * sizeof(T)));
    c->allocs.push_back(d);
    ACX_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<T*>(d);
    return ACX_OK;
}

static uint16_t to_bf16(float f) {      // round to nearest even, as v_cvt_pk_bf16_f32 does
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// rows x K fp32 -> rows x Kp bf16, source column k of group q (K = groups * Kg) lands at q * Kgp + k
static std::vector<uint16_t> bf16_rows(const std::vector<float>& w, int rows, int groups, int Kg, int Kgp) {
    std::vector<uint16_t> h((size_t)rows * groups * Kgp, 0);
    for (int n = 0; n < rows; ++n)
        for (int q = 0; q < groups; ++q)
            for (int k = 0; k < Kg; ++k)
                h[((size_t)n * groups + q) * Kgp + k] = to_bf16(w[((size_t)n * groups + q) * Kg + k]);
    return h;
}

// S16 form of gemm_split.hip: [rows][K/8][hi x8 | lo x8] fp16, values pre-multiplied by `scale` (a power of two)
static std::vector<uint16_t> s16_rows(const std::vector<float>& w, int rows, int K, float scale) {
    std::vector<uint16_t> h((size_t)rows * K * 2);
    for (int n = 0; n < rows; ++n)
        for (int k = 0; k < K; ++k) {
            const float v = w[(size_t)n * K + k] * scale;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            uint16_t* blk = h.data() + ((size_t)n * K + (size_t)(k & ~7)) * 2;
            std::memcpy(blk + (k & 7), &hi, 2);
            std::memcpy(blk + 8 + (k & 7), &lo, 2);
        }
    return h;
}

// power of two that brings max |w| into [2^14, 2^15): both fp16 halves of every weight within 2^13 of the largest
// stay normal numbers
static float s16_scale(const std::vector<float>& w) {
    float mx = 0.f;
    for (float v : w) mx = std::fmax(mx, std::fabs(v));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 1.f;
    return std::ldexp(1.0f, 14 - std::ilogb(mx));
}

// Power-of-two scale of the S16 hidden activation of one block.  The GEMM input is z = LayerNorm(y) without affine
// (the affine is folded into w1 / b1): ||z||_2 <= sqrt(C), so |h_n| = |w1_n . z + b1_n| <= ||w1_n||_2 sqrt(C) + |b1_n|
// (Cauchy-Schwarz) and |GELU(h)| <= |h|.  The scale puts that bound below the largest fp16 number: the GELU's
// fp32 -> fp16 conversion (split_math.h) cannot overflow, whatever the input (bounded below at 2^-24 -- weights of ~1e11).  Typical weights give 2^10..2^11; the absolute
// resolution of a stored value is 2^-25 / scale (fp16 subnormal spacing of the lo half).
static float hidden_scale_for(const std::vector<float>& w1, const std::vector<float>& b1, int N, int C) {
    double worst = 0.0;
    for (int n = 0; n < N; ++n) {
        double ss = 0.0;
        for (int k = 0; k < C; ++k) ss += (double)w1[(size_t)n * C + k] * w1[(size_t)n * C + k];
        worst = std::fmax(worst, std::sqrt(ss) * std::sqrt((double)C) + std::fabs((double)b1[n]));
    }
    if (!(worst > 0.0) || !std::isfinite(worst)) return 1.f;
    int e = (int)std::floor(std::log2(65000.0 / worst));
    e = e > 12 ? 12 : (e < -24 ? -24 : e);      // 2^-24: every scaled GELU coefficient stays a normal fp32 number (split_math.h, gelu_k3)
    return std::ldexp(1.0f, e);
}

// bf16 MFMA operands (both bf16 modes); bf16 activations in HBM for the stages that keep them (ACX_PREC_BF16_ACT, stages 0-2)
static bool is_bf16(const acx_ctx* c) { return c->precision == ACX_PREC_BF16 || c->precision == ACX_PREC_BF16_ACT; }
static bool act_bf16(const acx_ctx* c, int stage) { return c->precision == ACX_PREC_BF16_ACT && stage >= 0 && stage < 3; }

static void free_device(acx_ctx* c) {
    for (void* p : c->allocs) (void)hipFree(p);
    c->allocs.clear();
    for (int s = 0; s < 4; ++s) c->blocks[s].clear();
    c->d_dw_sink = nullptr;
    c->finalized = false;
}

static const std::vector<float>& W(const acx_ctx* c, const std::string& k) { return c->host.at(k).data; }

// max |stored - window x DFT| with the window read from bin 0 (cos = 1: that row IS the window)
static double stft_deviation_from_dft(const acx_ctx* c, std::vector<float>* hann_out) {
    const auto& re = W(c, "spectrogram_extractor.stft.conv_real.weight");
    const auto& im = W(c, "spectrogram_extractor.stft.conv_imag.weight");
    std::vector<float>& hann = *hann_out;
    hann.assign(re.begin(), re.begin() + kNFFT);      // bin 0: cos = 1, so the row IS the window
    double worst = 0.0;
    for (int k = 0; k < kBins; ++k) {
        for (int n = 0; n < kNFFT; ++n) {
            const double ang = 2.0 * M_PI * (double)(((long long)n * k) % kNFFT) / kNFFT;
            const double er = hann[n] * std::cos(ang), ei = -(double)hann[n] * std::sin(ang);
            worst = std::fmax(worst, std::fabs(er - re[(size_t)k * kNFFT + n]));
            worst = std::fmax(worst, std::fabs(ei - im[(size_t)k * kNFFT + n]));
        }
    }
    return worst;
}

static int finalize_impl(acx_ctx* c) {
    for (const auto& ks : key_table()) {
        auto it = c->host.find(ks.key);
        if (it == c->host.end()) ACX_FAIL(ACX_ERR_STATE, "missing weight '%s'", ks.key.c_str());
    }
    ACX_HIP(hipSetDevice(c->device));
    free_device(c);
    {
        // the two-GEMM form of stages 0-1 exists in the native fp32 arithmetic only (the 16-bit GEMMs have no N = 96 tile)
        const char* e = std::getenv("ACX_DISABLE_FUSED_MLP");
        c->use_fused_mlp = !(e && e[0] == '1' && c->precision == ACX_PREC_F32);
    }

    // ---- frontend ---------------------------------------------------------------------------
    // The FFT stands in for the two Conv1d only if the stored buffers ARE window x DFT (any window; torchlibrosa's is the
    // periodic hann).  The reference applies whatever its state_dict holds (convnext.py:179-187, overwritten by
    // load_state_dict), so anything else -- a fine-tuned or hand-edited frontend -- runs as the dense contraction it is:
    // frames [B T, 1024] . [conv_real; conv_imag]^T on the f32 matrix cores (frontend.hip).
    std::vector<float> hann;
    const double dev = stft_deviation_from_dft(c, &hann);
    c->stft_deviation = (float)dev;
    c->dense_stft = c->force_dense_stft || !(dev <= 2e-6);
    c->d_stft_w = nullptr; c->d_stft_zero = nullptr;
    if (c->dense_stft) {
        std::vector<float> w((size_t)kDenseN * kNFFT, 0.f);
        const auto& re = W(c, "spectrogram_extractor.stft.conv_real.weight");
        const auto& im = W(c, "spectrogram_extractor.stft.conv_imag.weight");
        std::memcpy(w.data(), re.data(), (size_t)kBins * kNFFT * 4);
        std::memcpy(w.data() + (size_t)kBins * kNFFT, im.data(), (size_t)kBins * kNFFT * 4);
        ACX_TRY(upload(c, w, &c->d_stft_w));
        ACX_TRY(upload(c, std::vector<float>(kDenseN, 0.f), &c->d_stft_zero));
    }
    ACX_TRY(upload(c, hann, &c->d_hann));
    std::vector<float> tw(2 * kNFFT);
    for (int n = 0; n < kNFFT; ++n) {
        const double a = -2.0 * M_PI * n / kNFFT;
        tw[2 * n] = (float)std::cos(a);
        tw[2 * n + 1] = (float)std::sin(a);
    }
    ACX_TRY(upload(c, tw, &c->d_twiddle));
    {
        const auto& melW = W(c, "logmel_extractor.melW");     // [513][224]
        std::vector<int> start(kMels), len(kMels), off(kMels);
        std::vector<float> band;
        for (int m = 0; m < kMels; ++m) {
            int lo = -1, hi = -1;
            for (int k = 0; k < kBins; ++k)
                if (melW[(size_t)k * kMels + m] != 0.f) { if (lo < 0) lo = k; hi = k; }
            start[m] = lo < 0 ? 0 : lo;
            len[m] = lo < 0 ? 0 : hi - lo + 1;
            off[m] = (int)band.size();
            for (int k = 0; k < len[m]; ++k) band.push_back(melW[(size_t)(start[m] + k) * kMels + m]);
        }
        if (band.empty()) band.push_back(0.f);
        ACX_TRY(upload(c, start, &c->d_mel_start));
        ACX_TRY(upload(c, len, &c->d_mel_len));
        ACX_TRY(upload(c, off, &c->d_mel_off));
        ACX_TRY(upload(c, band, &c->d_mel_w));
        c->mel_w_len = (int)band.size();
    }
    {
        const auto &w = W(c, "bn0.weight"), &b = W(c, "bn0.bias"), &mu = W(c, "bn0.running_mean"),
                   &var = W(c, "bn0.running_var");
        std::vector<float> sc(kMels), sh(kMels);
        for (int m = 0; m < kMels; ++m) {
            const double s = (double)w[m] / std::sqrt((double)var[m] + 1e-5);   // BatchNorm2d eps default
            sc[m] = (float)s;
            sh[m] = (float)((double)b[m] - (double)mu[m] * s);
        }
        ACX_TRY(upload(c, sc, &c->d_bn_scale));
        ACX_TRY(upload(c, sh, &c->d_bn_shift));
        ACX_TRY(upload(c, std::vector<float>(kMels, 1.f), &c->d_bn_one));
        ACX_TRY(upload(c, std::vector<float>(kMels, 0.f), &c->d_bn_zero));
    }
    {   // where the column-streaming depthwise kernel parks the stores of rows that are not image rows (dwconv_col.hip)
        void* d = nullptr;
        ACX_HIP(hipMalloc(&d, kDwSinkBytes));
        c->allocs.push_back(d);
        c->d_dw_sink = d;
    }
    // ---- stem -------------------------------------------------------------------------------
    ACX_TRY(upload(c, W(c, "downsample_layers.0.0.weight"), &c->d_stem_w));
    ACX_TRY(upload(c, W(c, "downsample_layers.0.0.bias"), &c->d_stem_b));
    ACX_TRY(upload(c, W(c, "downsample_layers.0.1.weight"), &c->d_stem_lnw));
    ACX_TRY(upload(c, W(c, "downsample_layers.0.1.bias"), &c->d_stem_lnb));
    // ---- downsample convs: LayerNorm affine folded, K ordered (dy,dx,c) ------------------------
    char buf[96];
    for (int i = 1; i < 4; ++i) {
        const int Ci = kDims[i - 1], Co = kDims[i];
        auto key = [&](const char* leaf) { snprintf(buf, sizeof buf, "downsample_layers.%d.%s", i, leaf); return std::string(buf); };
        const auto &lnw = W(c, key("0.weight")), &lnb = W(c, key("0.bias")), &cw = W(c, key("1.weight")),
                   &cb = W(c, key("1.bias"));
        std::vector<float> w((size_t)Co * 4 * Ci), b(Co);
        for (int n = 0; n < Co; ++n) {
            double acc = cb[n];
            for (int ci = 0; ci < Ci; ++ci)
                for (int q = 0; q < 4; ++q) {
                    const float v = cw[(((size_t)n * Ci + ci) * 2 + (q >> 1)) * 2 + (q & 1)];
                    w[(size_t)n * 4 * Ci + (size_t)q * Ci + ci] = (float)((double)v * lnw[ci]);
                    acc += (double)v * lnb[ci];
                }
            b[n] = (float)acc;
        }
        ACX_TRY(upload(c, w, &c->down[i].w));
        ACX_TRY(upload(c, b, &c->down[i].b));
        c->down[i].wh = nullptr;
        if (is_bf16(c)) ACX_TRY(upload(c, bf16_rows(w, Co, 4, Ci, pad64(Ci)), &c->down[i].wh));
        c->down[i].ws = nullptr;
        if (c->precision == ACX_PREC_F32_SPLIT) {
            c->down[i].ws_scale = s16_scale(w);
            ACX_TRY(upload(c, s16_rows(w, Co, 4 * Ci, c->down[i].ws_scale), &c->down[i].ws));
        }
    }
    // ---- blocks -----------------------------------------------------------------------------
    for (int s = 0; s < 4; ++s) {
        const int C = kDims[s];
        for (int j = 0; j < kDepths[s]; ++j) {
            auto key = [&](const char* leaf) { snprintf(buf, sizeof buf, "stages.%d.%d.%s", s, j, leaf); return std::string(buf); };
            const auto &gamma = W(c, key("gamma")), &dw = W(c, key("dwconv.weight")), &dwb = W(c, key("dwconv.bias")),
                       &lnw = W(c, key("norm.weight")), &lnb = W(c, key("norm.bias")), &w1 = W(c, key("pwconv1.weight")),
                       &b1 = W(c, key("pwconv1.bias")), &w2 = W(c, key("pwconv2.weight")), &b2 = W(c, key("pwconv2.bias"));
            BlockW bw;
            std::vector<float> t((size_t)49 * C);
            for (int ch = 0; ch < C; ++ch)
                for (int tap = 0; tap < 49; ++tap) t[(size_t)tap * C + ch] = dw[(size_t)ch * 49 + tap];
            ACX_TRY(upload(c, t, &bw.dw));
            ACX_TRY(upload(c, dwb, &bw.dwb));
            if (c->precision == ACX_PREC_BF16_ACT && s < 3) {
                // dwconv_mfma.hip: the B operands of the 16-block 4x4x4 MFMA, one 8-byte load per lane and operand.  Operand
                // (kernel row kh, d = input quad - output quad + 1, channel set) of lane (q = lane & 3, cl = lane >> 2) holds, for
                // k = 0..3, the weight of tap 4 d + k - q - 1 of row kh (zero outside 0..6) of channel 32 slice + 2 cl + set.
                std::vector<uint16_t> ops((size_t)(C / 32) * 42 * 64 * 4);
                for (int sl = 0; sl < C / 32; ++sl)
                    for (int kh = 0; kh < 7; ++kh)
                        for (int d = 0; d < 3; ++d)
                            for (int st = 0; st < 2; ++st)
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int q = lane & 3, ch = 32 * sl + 2 * (lane >> 2) + st;
                                    for (int k = 0; k < 4; ++k) {
                                        const int tp = 4 * d + k - q - 1;
                                        ops[((((size_t)(sl * 7 + kh) * 3 + d) * 2 + st) * 64 + lane) * 4 + k] =
                                            (tp >= 0 && tp < 7) ? to_bf16(dw[(size_t)ch * 49 + kh * 7 + tp]) : (uint16_t)0;
                                    }
                                }
                ACX_TRY(upload(c, ops, &bw.dw_ops));
            }
            std::vector<float> f1((size_t)4 * C * C), fb1((size_t)4 * C), fs1((size_t)4 * C);
            for (int n = 0; n < 4 * C; ++n) {
                double acc = b1[n], csum = 0.0;
                for (int k = 0; k < C; ++k) {
                    const float v = w1[(size_t)n * C + k];
                    const float f = (float)((double)v * lnw[k]);
                    f1[(size_t)n * C + k] = f;
                    csum += (double)f;                      // of the fp32 values the GEMM really multiplies
                    acc += (double)v * lnb[k];
                }
                fb1[n] = (float)acc;
                fs1[n] = (float)csum;
            }
            ACX_TRY(upload(c, f1, &bw.w1));
            ACX_TRY(upload(c, fb1, &bw.b1));
            ACX_TRY(upload(c, fs1, &bw.w1sum));
            std::vector<float> f2((size_t)C * 4 * C), fb2(C);
            for (int n = 0; n < C; ++n) {
                for (int k = 0; k < 4 * C; ++k) f2[(size_t)n * 4 * C + k] = (float)((double)gamma[n] * w2[(size_t)n * 4 * C + k]);
                fb2[n] = (float)((double)gamma[n] * b2[n]);
            }
            ACX_TRY(upload(c, f2, &bw.w2));
            ACX_TRY(upload(c, fb2, &bw.b2));
            if (is_bf16(c)) {
                ACX_TRY(upload(c, bf16_rows(f1, 4 * C, 1, C, pad64(C)), &bw.w1h));
                ACX_TRY(upload(c, bf16_rows(f2, C, 1, 4 * C, 4 * C), &bw.w2h));
                if (c->use_fused_mlp && mlp_fused_wide_bf16_supported(C)) {
                    // mlp_fused_wide_bf16.hip: one stream of 128 C-byte segments in consumption order
                    //   W1(0) W1(1) W2(0) W1(2) W2(1) ... W1(n-1) W2(n-2) W2(n-1),   n = 4C/64 chunks of 64 hidden units,
                    // each already in LDS image order.
                    const int nch = 4 * C / 64;
                    const size_t seg = (size_t)64 * C;                   // uint16 elements per segment
                    std::vector<uint16_t> st((size_t)2 * nch * seg);
                    for (int k = 0; k < nch; ++k) {
                        // W1 image: row r = hidden unit 64k + r (2 C bytes = C/8 chunks of 8 channels), chunk p at p ^ swz(r)
                        uint16_t* w1img = st.data() + (size_t)(k == 0 ? 0 : 2 * k - 1) * seg;
                        for (int r = 0; r < 64; ++r)
                            for (int p = 0; p < C / 8; ++p) {
                                const int pos = p ^ mlp_fused_wide_bf16_swz(C, r);
                                for (int e = 0; e < 8; ++e)
                                    w1img[(size_t)r * C + (size_t)pos * 8 + e] = to_bf16(0.5f * f1[(size_t)(64 * k + r) * C + 8 * p + e]);   // 0.5 W1: the accumulator is z (gelu2h_micro)
                            }
                        // W2 image: row = out channel (128 B = 8 chunks); chunk b = 2 s' + h (s' = k-step 0..3) holds hidden
                        // units 64k + 32(s' >> 1) + 16(s' & 1) + 4h + 8(jj >> 2) + (jj & 3), at position b ^ ((ch >> 1) & 7)
                        uint16_t* w2img = st.data() + (size_t)(k == nch - 1 ? 2 * nch - 1 : 2 * k + 2) * seg;
                        for (int ch = 0; ch < C; ++ch)
                            for (int b = 0; b < 8; ++b) {
                                const int sp = b >> 1, h = b & 1;
                                const int pos = b ^ ((ch >> 1) & 7);
                                for (int jj = 0; jj < 8; ++jj) {
                                    const int u = 64 * k + 32 * (sp >> 1) + 16 * (sp & 1) + 4 * h + 8 * (jj >> 2) + (jj & 3);
                                    w2img[(size_t)ch * 64 + (size_t)pos * 8 + jj] = to_bf16(f2[(size_t)ch * 4 * C + u]);
                                }
                            }
                    }
                    ACX_TRY(upload(c, st, &bw.wstream_b));
                }
            }
            if (c->precision == ACX_PREC_F32_SPLIT) {
                bw.w1s_scale = s16_scale(f1);
                bw.w2s_scale = s16_scale(f2);
                bw.hid_scale = hidden_scale_for(f1, fb1, 4 * C, C);
                const std::vector<uint16_t> h1 = s16_rows(f1, 4 * C, C, bw.w1s_scale);
                ACX_TRY(upload(c, h1, &bw.w1s));
                ACX_TRY(upload(c, s16_rows(f2, C, 4 * C, bw.w2s_scale), &bw.w2s));
                if (mlp_fused_split_supported(C)) {
                    // chunk-major image for mlp_fused_split.hip: per chunk j [W1c = rows 32j..32j+31 of w1s]
                    // [W2c: C rows x 4 blocks; block b = 2s'+h holds hidden units 32j + 16s' + 4h + 8(jj>>2) + (jj&3)]
                    const int nch = 4 * C / 32;
                    const size_t half = (size_t)64 * C;                    // uint16 elements per [32][C] S16 image
                    std::vector<uint16_t> pk((size_t)nch * 2 * half);
                    for (int j = 0; j < nch; ++j) {
                        uint16_t* blk = pk.data() + (size_t)j * 2 * half;
                        std::memcpy(blk, h1.data() + (size_t)j * half, half * 2);
                        uint16_t* blk2 = blk + half;
                        for (int ch = 0; ch < C; ++ch)
                            for (int b = 0; b < 4; ++b)
                                for (int jj = 0; jj < 8; ++jj) {
                                    const int u = 32 * j + 16 * (b >> 1) + 4 * (b & 1) + 8 * (jj >> 2) + (jj & 3);
                                    const float v = f2[(size_t)ch * 4 * C + u] * bw.w2s_scale;
                                    const _Float16 hi = (_Float16)v;
                                    const _Float16 lo = (_Float16)(v - (float)hi);
                                    std::memcpy(blk2 + (size_t)ch * 64 + b * 16 + jj, &hi, 2);
                                    std::memcpy(blk2 + (size_t)ch * 64 + b * 16 + 8 + jj, &lo, 2);
                                }
                    }
                    ACX_TRY(upload(c, pk, &bw.wpack_s));
                }
            }
            if (c->precision == ACX_PREC_F32_SPLIT && c->use_fused_mlp && mlp_fused_wide_supported(C)) {
                // mlp_fused_wide.hip: ONE stream of 128 C-byte segments in consumption order
                //   W1(0) W1(1) W2(0) W1(2) W2(1) ... W1(n-1) W2(n-2) W2(n-1),   n = 4C/32 hidden chunks,
                // each already in LDS image order (XOR swizzles baked in): a 1-KB LDS-DMA piece is 1 KB of the stream.
                const int nch = 4 * C / 32;
                const size_t seg = (size_t)64 * C;                   // uint16 elements per segment (128 C bytes)
                const std::vector<uint16_t> h1 = s16_rows(f1, 4 * C, C, bw.w1s_scale);      // [4C][C/8][hi8 | lo8]
                std::vector<uint16_t> st((size_t)2 * nch * seg);
                for (int k = 0; k < nch; ++k) {
                    // W1 image of chunk k: row r = hidden unit 32k + r (4 C bytes = C/4 chunks of 16 B: chunk p = block p >> 1
                    // of the row, p & 1: hi / lo halves), content chunk p at position p ^ swz(r): r & 15 when C/4 is a multiple
                    // of 16 chunks, (r >> 1) & 7 for C = 96 (24 chunks per row: odd rows start 8 chunks further)
                    uint16_t* w1img = st.data() + (size_t)(k == 0 ? 0 : 2 * k - 1) * seg;
                    for (int r = 0; r < 32; ++r)
                        for (int p = 0; p < C / 4; ++p) {
                            const int pos = p ^ ((C % 64 == 0) ? (r & 15) : ((r >> 1) & 7));
                            std::memcpy(w1img + ((size_t)r * 4 * C + (size_t)pos * 16) / 2,
                                        h1.data() + ((size_t)(32 * k + r) * C * 4 + (size_t)p * 16) / 2, 16);
                        }
                    // W2 image of chunk k: row = out channel (128 B = 8 chunks); content chunk 2b (hi) / 2b + 1 (lo) of
                    // block b (= k block g4 of the 16x16x32 MFMA) holds hidden units 32k + 16(jj >> 2) + 4b + (jj & 3) -- the order
                    // in which a lane's accumulators of phase 1 become its B operand of phase 2 --, at position ^ acx_swz8(ch)
                    uint16_t* w2img = st.data() + (size_t)(k == nch - 1 ? 2 * nch - 1 : 2 * k + 2) * seg;
                    for (int ch = 0; ch < C; ++ch)
                        for (int b = 0; b < 4; ++b) {
                            uint16_t hi8[8], lo8[8];
                            for (int jj = 0; jj < 8; ++jj) {
                                const int u = 32 * k + 16 * (jj >> 2) + 4 * b + (jj & 3);
                                const float v = f2[(size_t)ch * 4 * C + u] * bw.w2s_scale;
                                const _Float16 hi = (_Float16)v;
                                const _Float16 lo = (_Float16)(v - (float)hi);
                                std::memcpy(&hi8[jj], &hi, 2);
                                std::memcpy(&lo8[jj], &lo, 2);
                            }
                            const int sw = acx_swz8(ch);
                            std::memcpy(w2img + ((size_t)ch * 128 + (size_t)((2 * b) ^ sw) * 16) / 2, hi8, 16);
                            std::memcpy(w2img + ((size_t)ch * 128 + (size_t)((2 * b + 1) ^ sw) * 16) / 2, lo8, 16);
                        }
                }
                ACX_TRY(upload(c, st, &bw.wstream_s));
            }
            if (mlp_fused_supported(C)) {       // chunk-major image for the fused kernel's LDS-DMA
                const int nch = 4 * C / 32;
                std::vector<float> pk((size_t)nch * 64 * C);
                for (int j = 0; j < nch; ++j) {
                    float* blk = pk.data() + (size_t)j * 64 * C;
                    for (int r = 0; r < 32; ++r)
                        for (int k = 0; k < C; ++k) blk[(size_t)r * C + k] = f1[(size_t)(32 * j + r) * C + k];
                    float* blk2 = blk + (size_t)32 * C;
                    for (int ch = 0; ch < C; ++ch)
                        for (int h = 0; h < 32; ++h) blk2[(size_t)ch * 32 + h] = f2[(size_t)ch * 4 * C + 32 * j + h];
                }
                ACX_TRY(upload(c, pk, &bw.wpack));
            }
            c->blocks[s].push_back(bw);
        }
    }
    // ---- tail -------------------------------------------------------------------------------
    ACX_TRY(upload(c, W(c, "norm.weight"), &c->d_norm_w));
    ACX_TRY(upload(c, W(c, "norm.bias"), &c->d_norm_b));
    ACX_TRY(upload(c, W(c, "head_audioset.weight"), &c->d_head_w));
    ACX_TRY(upload(c, W(c, "head_audioset.bias"), &c->d_head_b));
    c->finalized = true;
    return ACX_OK;
}

static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }


struct Plan {          // workspace carve-up for one forward
    int T, Hs[4], Ws[4];
    size_t off_feat, off_x[4], off_y, off_hidden, off_stats, total;
};

static int make_plan(int B, int64_t L, Plan* p) {
    if (B <= 0) ACX_FAIL(ACX_ERR_ARG, "batch size must be positive (got %d)", B);
    if (L < ACX_MIN_SAMPLES)
        ACX_FAIL(ACX_ERR_SHAPE,
                 "clip of %lld samples is too short: the last 2x2 downsample needs at least %d samples "
                 "(kernel size can't be greater than actual input size)", (long long)L, ACX_MIN_SAMPLES);
    p->T = (int)(L / kHop + 1);
    p->Hs[0] = stage_h0(p->T);
    p->Ws[0] = kStemW;
    for (int s = 1; s < 4; ++s) { p->Hs[s] = p->Hs[s - 1] / 2; p->Ws[s] = p->Ws[s - 1] / 2; }
    size_t off = 0;
    p->off_feat = off; off += align_up((size_t)B * p->T * kMels * 4);
    for (int s = 0; s < 4; ++s) { p->off_x[s] = off; off += align_up((size_t)B * p->Hs[s] * p->Ws[s] * kDims[s] * 4); }
    const size_t pix0 = (size_t)B * p->Hs[0] * p->Ws[0];
    p->off_y = off; off += align_up(pix0 * kDims[0] * 4);
    p->off_hidden = off; off += align_up(pix0 * 4 * kDims[0] * 4);
    p->off_stats = off; off += align_up(pix0 * 2 * 4);
    p->total = off;
    return ACX_OK;
}

// bf16 precision: y -> fp32 LayerNorm -> bf16 rows; pwconv1 + GELU -> bf16 hidden; pwconv2 + residual -> fp32 x.
// Both bf16 arrays live in the `hidden` scratch (sized for the fp32 hidden activation): [M][4C] then [M][Cp].
static int run_mlp_bf16(acx_ctx* c, const BlockW& bw, int C, const float* y, float* x, float* hidden, int64_t M,
                        hipStream_t st) {
    const int Cp = pad64(C);
    char* hb = reinterpret_cast<char*>(hidden);
    char* yb = hb + align_up((size_t)M * 4 * C * 2);
    ACX_TRY(launch_layernorm_rows_bf16(c, y, yb, M, C, st));
    GemmBf16Args g1{};
    g1.A = yb; g1.Wt = bw.w1h; g1.bias = bw.b1; g1.out = hb; g1.M = M; g1.N = 4 * C; g1.Kp = Cp; g1.lda = Cp;
    g1.epi = EPI_GELU; g1.cls = ACX_K_PW1;
    ACX_TRY(launch_gemm_bf16(c, g1, st));
    GemmBf16Args g2{};
    g2.A = hb; g2.Wt = bw.w2h; g2.bias = bw.b2; g2.out = x; g2.resid = x; g2.M = M; g2.N = C; g2.Kp = 4 * C; g2.lda = 4 * C;
    g2.epi = EPI_RESID; g2.cls = ACX_K_PW2;
    return launch_gemm_bf16(c, g2, st);
}

// split precision: y -> fp32 LayerNorm -> S16 rows IN PLACE (row-local); pwconv1 + GELU -> S16 hidden;
// pwconv2 + residual -> fp32 x.  Same bytes per element as the fp32 path.
static int run_mlp_split(acx_ctx* c, const BlockW& bw, int C, float* y, float* x, float* hidden, int64_t M,
                         hipStream_t st) {
    ACX_TRY(launch_layernorm_rows_split(c, y, y, M, C, st));
    GemmSplitArgs g1{};
    g1.A = y; g1.Wt = bw.w1s; g1.bias = bw.b1; g1.out = hidden; g1.M = M; g1.N = 4 * C; g1.K = C;
    g1.sinv = 1.0f / (kSplitLnScale * bw.w1s_scale); g1.hscale = bw.hid_scale; g1.epi = EPI_GELU; g1.cls = ACX_K_PW1;
    ACX_TRY(launch_gemm_split(c, g1, st));
    GemmSplitArgs g2{};
    g2.A = hidden; g2.Wt = bw.w2s; g2.bias = bw.b2; g2.out = x; g2.resid = x; g2.M = M; g2.N = C; g2.K = 4 * C;
    g2.sinv = 1.0f / (bw.hid_scale * bw.w2s_scale); g2.epi = EPI_RESID; g2.cls = ACX_K_PW2;
    return launch_gemm_split(c, g2, st);
}

// True when the last block of stage s can hand the downsample conv its LayerNorm'ed S16 operand directly
// (fused split MLP kernel, LNOUT epilogue): x of that stage is then NOT updated by its last block.
static bool block_can_emit_ln(const acx_ctx* c, int s) {
    if (s < 3 && is_bf16(c) && c->use_fused_mlp && mlp_fused_wide_bf16_supported(kDims[s])) return true;
    return s < 3 && c->precision == ACX_PREC_F32_SPLIT && c->use_fused_mlp &&
           (mlp_fused_split_supported(kDims[s]) || mlp_fused_wide_supported(kDims[s]));
}

static int run_block(acx_ctx* c, int s, int j, float* x, float* y, float* hidden, float* stats, int B, int H, int Wd,
                     hipStream_t st, void* ln_out = nullptr) {
    const int C = kDims[s];
    const BlockW& bw = c->blocks[s][j];
    const int64_t M = (int64_t)B * H * Wd;
    if (c->precision == ACX_PREC_F32_SPLIT) {
        ACX_TRY(launch_dwconv(c, bw, C, x, y, nullptr, B, H, Wd, st));
        if (c->use_fused_mlp && mlp_fused_wide_supported(C) && bw.wstream_s) return launch_mlp_fused_wide(c, bw, C, y, x, M, st, ln_out);
        if (c->use_fused_mlp && mlp_fused_split_supported(C)) return launch_mlp_fused_split(c, bw, C, y, x, M, st, ln_out);
        if (ln_out) ACX_FAIL(ACX_ERR_STATE, "run_block: LayerNorm output requested from a two-GEMM stage");
        return run_mlp_split(c, bw, C, y, x, hidden, M, st);
    }
    if (is_bf16(c)) {
        const bool ab = act_bf16(c, s);         // x and y of this stage are bf16 tensors in HBM
        ACX_TRY(launch_dwconv(c, bw, C, x, y, nullptr, B, H, Wd, st, ab));
        if (c->use_fused_mlp && bw.wstream_b) return launch_mlp_fused_wide_bf16(c, bw, C, y, x, M, st, ln_out, pad64(C), ab);
        if (ab) ACX_FAIL(ACX_ERR_STATE, "run_block: bf16 activations need the fused block kernel (stage %d)", s);
        if (ln_out) ACX_FAIL(ACX_ERR_STATE, "run_block: LayerNorm output requested from a two-GEMM stage");
        return run_mlp_bf16(c, bw, C, y, x, hidden, M, st);
    }
    if (c->use_fused_mlp && mlp_fused_supported(C)) {
        ACX_TRY(launch_dwconv(c, bw, C, x, y, nullptr, B, H, Wd, st));      // LN statistics are computed in-kernel
        return launch_mlp_fused(c, bw, C, y, x, M, st);
    }
    ACX_TRY(launch_dwconv(c, bw, C, x, y, stats, B, H, Wd, st));
    GemmArgs g1{};
    g1.A = y; g1.Wt = bw.w1; g1.bias = bw.b1; g1.out = hidden; g1.stats = stats; g1.colsum = bw.w1sum; g1.M = M; g1.N = 4 * C; g1.K = C;
    g1.epi = EPI_GELU; g1.cls = ACX_K_PW1;
    ACX_TRY(launch_gemm(c, g1, st));
    GemmArgs g2{};
    g2.A = hidden; g2.Wt = bw.w2; g2.bias = bw.b2; g2.out = x; g2.resid = x; g2.M = M; g2.N = C; g2.K = 4 * C;
    g2.epi = EPI_RESID; g2.cls = ACX_K_PW2;
    ACX_TRY(launch_gemm(c, g2, st));
    return ACX_OK;
}

// have_ln: xnorm already holds the normalised S16 rows (written by the last block of the previous stage)
// out_bf16: the result is the bf16 activation tensor of stage i (ACX_PREC_BF16_ACT inside acx_forward; the per-layer entry
// point keeps fp32)
static int run_downsample(acx_ctx* c, int i, const float* x, float* out, float* xnorm, int B, int H, int Wd,
                          hipStream_t st, bool have_ln = false, bool out_bf16 = false) {
    const int Ci = kDims[i - 1], Co = kDims[i];
    if (c->precision == ACX_PREC_F32_SPLIT) {
        if (!have_ln) ACX_TRY(launch_layernorm_rows_split(c, x, xnorm, (int64_t)B * H * Wd, Ci, st));
        GemmSplitArgs g{};
        g.A = xnorm; g.Wt = c->down[i].ws; g.bias = c->down[i].b; g.out = out;
        g.gather = 1; g.H = H; g.W = Wd; g.C = Ci; g.Ho = H / 2; g.Wo = Wd / 2;
        g.M = (int64_t)B * g.Ho * g.Wo; g.N = Co; g.K = 4 * Ci; g.sinv = 1.0f / (kSplitLnScale * c->down[i].ws_scale);
        g.epi = EPI_BIAS; g.cls = ACX_K_DOWNSAMPLE;
        return launch_gemm_split(c, g, st);
    }
    if (is_bf16(c)) {
        const int Cp = pad64(Ci);
        if (!have_ln) ACX_TRY(launch_layernorm_rows_bf16(c, x, xnorm, (int64_t)B * H * Wd, Ci, st));
        GemmBf16Args g{};
        g.out_bf16 = out_bf16 ? 1 : 0;
        g.A = xnorm; g.Wt = c->down[i].wh; g.bias = c->down[i].b; g.out = out;
        g.gather = 1; g.H = H; g.W = Wd; g.Cp = Cp; g.Ho = H / 2; g.Wo = Wd / 2;
        g.M = (int64_t)B * g.Ho * g.Wo; g.N = Co; g.Kp = 4 * Cp; g.lda = Cp; g.epi = EPI_BIAS; g.cls = ACX_K_DOWNSAMPLE;
        return launch_gemm_bf16(c, g, st);
    }
    ACX_TRY(launch_layernorm_rows(c, x, xnorm, (int64_t)B * H * Wd, Ci, st));
    GemmArgs g{};
    g.A = xnorm; g.Wt = c->down[i].w; g.bias = c->down[i].b; g.out = out;
    g.gather = 1; g.H = H; g.W = Wd; g.C = Ci; g.Ho = H / 2; g.Wo = Wd / 2;
    g.M = (int64_t)B * g.Ho * g.Wo; g.N = Co; g.K = 4 * Ci; g.epi = EPI_BIAS; g.cls = ACX_K_DOWNSAMPLE;
    return launch_gemm(c, g, st);
}

static int need_ready(const acx_ctx* c) {
    if (!c) ACX_FAIL(ACX_ERR_ARG, "null context");
    if (!c->finalized) ACX_FAIL(ACX_ERR_STATE, "weights not finalized: call acx_set_weight for every key, then acx_finalize");
    return ACX_OK;
}

static int check_stage(int stage, int block) {
    if (stage < 0 || stage > 3 || block < 0 || block >= kDepths[stage])
        ACX_FAIL(ACX_ERR_ARG, "no block %d in stage %d", block, stage);
    return ACX_OK;
}

}  // namespace acx

using namespace acx;

// side streams + fork/join events of the batch split (acx_forward), one set per caller stream.  The set of the null stream is
// created in acx_create; any other stream gets its set at its first split forward -- which must not be a stream capture
// (hipStreamCreate inside a capture region): warm up once on the stream before capturing, as the host wrapper does.
// A partial failure destroys what it created; the map is bounded (sets of the least recently added streams are dropped).
static void destroy_aux(acx_ctx::Aux& a) {
    if (a.fork) (void)hipEventDestroy(a.fork);
    for (auto& j : a.joins) if (j) (void)hipEventDestroy(j);
    for (auto& s : a.streams) if (s) (void)hipStreamDestroy(s);
    a = acx_ctx::Aux{};
}
static int make_aux(acx_ctx::Aux* a) {
    *a = acx_ctx::Aux{};
    hipError_t e = hipEventCreateWithFlags(&a->fork, hipEventDisableTiming);
    for (int i = 0; i < acx_ctx::kMaxSplitWays - 1 && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&a->streams[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&a->joins[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        destroy_aux(*a);
        ACX_FAIL(ACX_ERR_HIP, "side streams / events of the batch split: %s", hipGetErrorString(e));
    }
    return ACX_OK;
}
extern "C" {

const char* acx_last_error(void) { return g_err; }
int acx_version(void) { return 100; }

int acx_create(int hip_device, acx_ctx** out) {
    if (!out) ACX_FAIL(ACX_ERR_ARG, "acx_create: out is null");
    int n = 0;
    ACX_HIP(hipGetDeviceCount(&n));
    if (hip_device < 0 || hip_device >= n) ACX_FAIL(ACX_ERR_ARG, "acx_create: device %d out of range (%d visible)", hip_device, n);
    hipDeviceProp_t prop;
    ACX_HIP(hipGetDeviceProperties(&prop, hip_device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        ACX_FAIL(ACX_ERR_UNSUPPORTED, "acx_create: device %d is %s; this library is built for gfx950 (MI355X) only",
                 hip_device, prop.gcnArchName);
    acx_ctx* c = new acx_ctx();
    c->device = hip_device;
    ACX_HIP(hipSetDevice(hip_device));
    {
        const char* e = std::getenv("ACX_SPLIT_STREAMS");
        c->split_streams = !(e && e[0] == '0');
        const char* e2 = std::getenv("ACX_SPLIT_WAYS");
        if (e2 && e2[0] >= '1' && e2[0] <= '0' + acx_ctx::kMaxSplitWays && e2[1] == 0) c->split_ways = e2[0] - '0';
    }
    {   // the null stream's side stream + events exist from the start (nothing is created inside a capture)
        acx_ctx::AuxEntry e;
        int rc = make_aux(&e.a);
        if (rc != ACX_OK) { delete c; return rc; }
        c->aux.emplace((hipStream_t) nullptr, e);
    }
    {
        static std::once_flag env_once;
        std::call_once(env_once, tuning_reload);
    }
    *out = c;
    return ACX_OK;
}

void acx_destroy(acx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_device(c);
    for (auto& r : c->prof.recs) { if (r.a) (void)hipEventDestroy(r.a); if (r.b) (void)hipEventDestroy(r.b); }
    for (auto e : c->prof.pool) (void)hipEventDestroy(e);
    for (auto& kv : c->aux) destroy_aux(kv.second.a);
    for (auto& a : c->aux_retired) destroy_aux(a);
    comm_release(c);
    delete c;
}

int acx_set_weight(acx_ctx* c, const char* key, const float* host_data, const int64_t* shape, int ndim) {
    if (!c || !key || !host_data || (ndim > 0 && !shape)) ACX_FAIL(ACX_ERR_ARG, "acx_set_weight: null argument");
    for (const auto& ks : key_table()) {
        if (ks.key != key) continue;
        if ((int)ks.shape.size() != ndim) ACX_FAIL(ACX_ERR_SHAPE, "'%s': expected %d dims, got %d", key, (int)ks.shape.size(), ndim);
        size_t n = 1;
        for (int d = 0; d < ndim; ++d) {
            if (shape[d] != ks.shape[d]) ACX_FAIL(ACX_ERR_SHAPE, "'%s': dim %d is %lld, expected %lld", key, d, (long long)shape[d], (long long)ks.shape[d]);
            n *= (size_t)shape[d];
        }
        HostTensor& t = c->host[key];
        t.shape.assign(shape, shape + ndim);
        t.data.assign(host_data, host_data + n);
        c->finalized = false;
        return ACX_OK;
    }
    ACX_FAIL(ACX_ERR_ARG, "acx_set_weight: unexpected key '%s'", key);
}

int acx_finalize(acx_ctx* c) {
    if (!c) ACX_FAIL(ACX_ERR_ARG, "null context");
    c->fail_sub.store(-1, std::memory_order_relaxed);       // an armed test hook never outlives the weights it was armed on
    int rc = finalize_impl(c);
    if (rc != ACX_OK) { free_device(c); }
    return rc;
}

int acx_set_precision(acx_ctx* c, int precision) {
    if (!c) ACX_FAIL(ACX_ERR_ARG, "null context");
    if (precision != ACX_PREC_F32 && precision != ACX_PREC_BF16 && precision != ACX_PREC_F32_SPLIT && precision != ACX_PREC_BF16_ACT) ACX_FAIL(ACX_ERR_ARG, "acx_set_precision: unknown precision %d", precision);
    if (precision != c->precision) { c->precision = precision; c->finalized = false; }
    return ACX_OK;
}

int acx_set_frontend(acx_ctx* c, int mode) {
    if (!c) ACX_FAIL(ACX_ERR_ARG, "null context");
    if (mode != ACX_FRONTEND_AUTO && mode != ACX_FRONTEND_DENSE) ACX_FAIL(ACX_ERR_ARG, "acx_set_frontend: unknown mode %d", mode);
    const bool dense = mode == ACX_FRONTEND_DENSE;
    if (dense == c->force_dense_stft) return ACX_OK;
    c->force_dense_stft = dense;
    return c->finalized ? acx_finalize(c) : ACX_OK;
}

int acx_num_frames(int64_t L, int* T) {
    if (!T || L < 0) ACX_FAIL(ACX_ERR_ARG, "acx_num_frames: bad argument");
    *T = (int)(L / kHop + 1);
    return ACX_OK;
}

int acx_stage_hw(int64_t L, int stage, int* H, int* Wd) {
    if (!H || !Wd || stage < 0 || stage > 3) ACX_FAIL(ACX_ERR_ARG, "acx_stage_hw: bad argument");
    Plan p;
    ACX_TRY(make_plan(1, L, &p));
    *H = p.Hs[stage]; *Wd = p.Ws[stage];
    return ACX_OK;
}

// A batch is split into sub-batches that run side by side on separate streams when every sub-batch keeps at least this
// many clips.
static constexpr int kSplitMinClips = 8;

// number of sub-batches a forward of B clips runs as
static int split_ways_for(const acx_ctx* c, int B) {
    int ways = 1;
    if (c && c->split_streams && !c->prof.on) {
        // defaults measured at B = 64 (profiles/r03_j_batch_split.txt): sub-batches on separate streams fill each other's
        // tail rounds and kernel boundaries -- also between CU-exclusive kernels, whose workgroups never share a CU but do
        // share the chip
        ways = c->split_ways > 0 ? c->split_ways : 2;
    }
    while (ways > 1 && B < ways * kSplitMinClips) --ways;
    return ways;
}
// clips of sub-batch i of `ways`
static int split_part(int B, int ways, int i) { return B / ways + (i < B % ways ? 1 : 0); }

int acx_workspace_bytes(const acx_ctx* c, int B, int64_t L, int mode, size_t* out_bytes) {
    if (!out_bytes || mode < 0 || mode > 2) ACX_FAIL(ACX_ERR_ARG, "acx_workspace_bytes: bad argument");
    Plan p;
    ACX_TRY(make_plan(B, L, &p));
    *out_bytes = p.total;
    // room for every way the forward may run (the split can be switched by precision, profiling or the environment)
    for (int ways = 2; ways <= acx_ctx::kMaxSplitWays && B >= ways * kSplitMinClips; ++ways) {
        size_t tot = 0;
        for (int i = 0; i < ways; ++i) {
            Plan pi;
            ACX_TRY(make_plan(split_part(B, ways, i), L, &pi));
            tot += pi.total;
        }
        if (tot > *out_bytes) *out_bytes = tot;
    }
    return ACX_OK;
}

int acx_sub_batches(const acx_ctx* c, int B, int* out) {
    if (!c || !out || B <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_sub_batches: bad argument");
    *out = split_ways_for(c, B);
    return ACX_OK;
}

static constexpr size_t kMaxAuxStreams = 16;
// The fork/join set of caller stream `st`, pinned until release_aux().  *found = false (and nothing pinned) when the stream
// has no set yet and one cannot be made now -- the first split forward on the stream happens inside a stream capture
// (hipStreamCreate is illegal there; `with torch.cuda.graph(g): model(x)` captures on a private stream nobody can warm
// up): the caller then runs the batch un-split, which gives the same bits.
static int get_aux(acx_ctx* c, hipStream_t st, acx_ctx::Aux* out, bool* found) {
    std::lock_guard<std::mutex> lock(c->aux_mutex);
    *found = false;
    auto it = c->aux.find(st);
    if (it == c->aux.end()) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (st != nullptr && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return ACX_OK;
        acx_ctx::AuxEntry e;
        if (c->aux.size() >= kMaxAuxStreams) {          // drop the least recently used set nobody is queueing on
            auto victim = c->aux.end();
            for (auto d = c->aux.begin(); d != c->aux.end(); ++d)
                if (d->first != nullptr && d->second.users == 0 && (victim == c->aux.end() || d->second.stamp < victim->second.stamp))
                    victim = d;
            if (victim != c->aux.end()) {
                // retired, not destroyed: destroying a stream that still has work means synchronising it, which stalls every
                // forward waiting for aux_mutex and is an illegal call while ANY stream of the process captures in global
                // mode.  The retired sets are a FREE POOL: fork and join are ordered by events alone,
                // so a set whose old work is still draining can serve a new caller stream as it is -- a context called from
                // ever-new streams (per-request streams, private capture streams) recycles kMaxAuxStreams + pool sets for ever
                // instead of leaking two streams and three events per eviction.
                c->aux_retired.push_back(victim->second.a);
                c->aux.erase(victim);
            }
        }
        if (!c->aux_retired.empty()) {
            e.a = c->aux_retired.back();
            c->aux_retired.pop_back();
        } else {
            ACX_TRY(make_aux(&e.a));
        }
        it = c->aux.emplace(st, e).first;
    }
    it->second.stamp = ++c->aux_clock;
    it->second.users += 1;
    *out = it->second.a;
    *found = true;
    return ACX_OK;
}
static void release_aux(acx_ctx* c, hipStream_t st) {
    std::lock_guard<std::mutex> lock(c->aux_mutex);
    auto it = c->aux.find(st);
    if (it != c->aux.end() && it->second.users > 0) it->second.users -= 1;
}

static int forward_one(acx_ctx* c, const float* wav, int B, int64_t L, int mode, float* out0, float* out1, char* ws,
                       const Plan& p, hipStream_t st) {
    float* feat = (float*)(ws + p.off_feat);
    float* x[4];
    for (int s = 0; s < 4; ++s) x[s] = (float*)(ws + p.off_x[s]);
    float* y = (float*)(ws + p.off_y);
    float* hidden = (float*)(ws + p.off_hidden);
    float* stats = (float*)(ws + p.off_stats);

    // (dense-DFT fallback: frames in `hidden`, spectrum in `y` -- both idle until the first block, both large enough:
    // T * 4096 <= H0 * 86016 and T * 4 kDenseN <= H0 * 21504 bytes per clip with H0 >= (T + 1) / 4)
    ACX_TRY(launch_logmel(c, wav, B, L, p.T, feat, true, st, hidden, y));
    ACX_TRY(launch_stem(c, feat, B, p.T, p.Hs[0], x[0], st, act_bf16(c, 0)));
    for (int s = 0; s < 4; ++s) {
        // The last block of stages 0-2 writes LayerNorm(x) as GEMM operand rows (S16 / bf16) instead of x: nothing else
        // reads that x (convnext.py:270-273).  The rows go to the `hidden` scratch (idle in the fused stages), never to
        // y: the block kernel reads y -- with a tile of look-ahead -- while it writes them, and bf16 rows are shorter
        // than the fp32 rows they would overwrite.
        if (s > 0) {
            const bool have_ln = block_can_emit_ln(c, s - 1);
            if (act_bf16(c, s - 1) && !have_ln) ACX_FAIL(ACX_ERR_STATE, "bf16 activations: stage %d must hand its LayerNorm rows to the downsample conv", s - 1);
            ACX_TRY(run_downsample(c, s, x[s - 1], x[s], have_ln ? hidden : y, B, p.Hs[s - 1], p.Ws[s - 1], st, have_ln, act_bf16(c, s)));
        }
        for (int j = 0; j < kDepths[s]; ++j) {
            void* ln_out = (j == kDepths[s] - 1 && block_can_emit_ln(c, s)) ? (void*)hidden : nullptr;
            ACX_TRY(run_block(c, s, j, x[s], y, hidden, stats, B, p.Hs[s], p.Ws[s], st, ln_out));
        }
    }
    if (mode == ACX_MODE_FRAME) return launch_nhwc_to_nchw(c, x[3], out0, B, p.Hs[3], p.Ws[3], kDims[3], st);
    if (mode == ACX_MODE_SCENE) return launch_pool_head(c, x[3], B, p.Hs[3], out0, nullptr, nullptr, st);
    return launch_pool_head(c, x[3], B, p.Hs[3], nullptr, out0, out1, st);
}

int acx_forward(acx_ctx* c, const float* wav, int B, int64_t L, int mode, float* out0, float* out1, void* workspace,
                size_t workspace_bytes, void* stream) {
    ACX_TRY(need_ready(c));
    if (!wav || !out0 || !workspace) ACX_FAIL(ACX_ERR_ARG, "acx_forward: null pointer");
    if (mode < 0 || mode > 2) ACX_FAIL(ACX_ERR_ARG, "acx_forward: bad mode %d", mode);
    if (mode == ACX_MODE_LOGITS && !out1) ACX_FAIL(ACX_ERR_ARG, "acx_forward: logits mode needs out1 (probs)");
    size_t need = 0;
    ACX_TRY(acx_workspace_bytes(c, B, L, mode, &need));
    if (workspace_bytes < need) ACX_FAIL(ACX_ERR_WORKSPACE, "workspace of %zu bytes is smaller than the %zu needed", workspace_bytes, need);
    if (((uintptr_t)workspace & 255) != 0) ACX_FAIL(ACX_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    // Clips are independent: a large batch runs as sub-batches on separate streams (fork/join with events, so the call
    // still looks like one unit of work on `stream` and stays graph-capturable).  Per-kernel event profiling runs
    // un-split to keep launch durations clean.  The kernels of the 16-bit arithmetics are CU-exclusive -- nothing
    // shares a CU with them -- but a second stream's workgroups take the CUs their tail rounds and launch boundaries
    // leave idle: +4 % (fp32_split) / +11 % (bf16a) at B = 64.
    const int ways = split_ways_for(c, B);
    acx_ctx::Aux aux;
    bool have_aux = false;
    if (ways > 1) ACX_TRY(get_aux(c, st, &aux, &have_aux));
    if (ways > 1 && have_aux) {
        // From here on every exit joins the side streams that were forked (an error in sub-batch i must not leave an
        // un-joined fork behind -- inside a stream capture that invalidates the capture, and the caller may free the
        // workspace the side streams still use) and unpins the set.
        int rc = ACX_OK;
        int forked = 0;                 // side streams that wait on the fork event so far
        hipError_t he = hipEventRecord(aux.fork, st);
        if (he != hipSuccess) { set_error("hipEventRecord(fork) failed: %s", hipGetErrorString(he)); rc = ACX_ERR_HIP; }
        size_t ws_off = 0;
        int b_off = 0;
        tls_inflight_ways = ways;
        for (int i = 0; i < ways && rc == ACX_OK; ++i) {
            const int Bi = split_part(B, ways, i);
            Plan pi;
            rc = make_plan(Bi, L, &pi);
            if (rc != ACX_OK) break;
            const size_t per_clip = mode == ACX_MODE_FRAME ? (size_t)kDims[3] * pi.Hs[3] * pi.Ws[3]
                                                           : (mode == ACX_MODE_SCENE ? (size_t)kDims[3] : (size_t)kClasses);
            hipStream_t si = i == 0 ? st : aux.streams[i - 1];
            if (i > 0) {
                he = hipStreamWaitEvent(si, aux.fork, 0);
                if (he != hipSuccess) { set_error("hipStreamWaitEvent(fork) failed: %s", hipGetErrorString(he)); rc = ACX_ERR_HIP; break; }
                forked = i;
            }
            rc = forward_one(c, wav + (size_t)b_off * L, Bi, L, mode, out0 + b_off * per_clip,
                             out1 ? out1 + b_off * per_clip : nullptr, ws + ws_off, pi, si);
            ws_off += pi.total;
            b_off += Bi;
            if (rc == ACX_OK && c->fail_sub.load(std::memory_order_relaxed) == i) {
                set_error("test hook acx_test_fail_sub: failure injected after sub-batch %d", i);
                rc = ACX_ERR_STATE;
            }
        }
        tls_inflight_ways = 1;
        for (int i = 1; i <= forked; ++i) {            // join whatever was forked, also after an error
            hipError_t e1 = hipEventRecord(aux.joins[i - 1], aux.streams[i - 1]);
            if (e1 == hipSuccess) e1 = hipStreamWaitEvent(st, aux.joins[i - 1], 0);
            if (e1 != hipSuccess && rc == ACX_OK) { set_error("joining sub-batch %d failed: %s", i, hipGetErrorString(e1)); rc = ACX_ERR_HIP; }
        }
        release_aux(c, st);
        return rc;
    }
    Plan p;
    ACX_TRY(make_plan(B, L, &p));
    return forward_one(c, wav, B, L, mode, out0, out1, ws, p, st);
}

int acx_logmel_bn0(acx_ctx* c, const float* wav, int B, int64_t L, float* out, int apply_bn0, void* stream) {
    ACX_TRY(need_ready(c));
    if (!wav || !out || B <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_logmel_bn0: bad argument");
    if (L < kNFFT / 2 + 1) ACX_FAIL(ACX_ERR_SHAPE, "reflect padding needs at least %d samples (got %lld)", kNFFT / 2 + 1, (long long)L);
    return launch_logmel(c, wav, B, L, (int)(L / kHop + 1), out, apply_bn0 != 0, (hipStream_t)stream);
}

int acx_stem_ln(acx_ctx* c, const float* in, int B, int T, float* out, void* stream) {
    ACX_TRY(need_ready(c));
    if (!in || !out || B <= 0 || T <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_stem_ln: bad argument");
    return launch_stem(c, in, B, T, stage_h0(T), out, (hipStream_t)stream);
}

int acx_dwconv7(acx_ctx* c, int stage, int block, const float* x, float* y, float* stats, int B, int H, int Wd, void* stream) {
    ACX_TRY(need_ready(c));
    ACX_TRY(check_stage(stage, block));
    if (!x || !y || B <= 0 || H <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_dwconv7: bad argument");
    if (Wd != (kStemW >> stage)) ACX_FAIL(ACX_ERR_SHAPE, "stage %d has width %d, got %d", stage, kStemW >> stage, Wd);
    return launch_dwconv(c, c->blocks[stage][block], kDims[stage], x, y, stats, B, H, Wd, (hipStream_t)stream);
}

int acx_dwconv7_bf16(acx_ctx* c, int stage, int block, const uint16_t* x, uint16_t* y, int B, int H, int Wd, void* stream) {
    ACX_TRY(need_ready(c));
    ACX_TRY(check_stage(stage, block));
    if (!x || !y || B <= 0 || H <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_dwconv7_bf16: bad argument");
    if (!act_bf16(c, stage)) ACX_FAIL(ACX_ERR_STATE, "acx_dwconv7_bf16: stage %d does not store bf16 activations at this precision", stage);
    if (Wd != (kStemW >> stage)) ACX_FAIL(ACX_ERR_SHAPE, "stage %d has width %d, got %d", stage, kStemW >> stage, Wd);
    return launch_dwconv(c, c->blocks[stage][block], kDims[stage], x, y, nullptr, B, H, Wd, (hipStream_t)stream, true);
}

int acx_block_scratch_bytes(int stage, int B, int H, int Wd, size_t* out_bytes) {
    if (!out_bytes || stage < 0 || stage > 3 || B <= 0 || H <= 0 || Wd <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_block_scratch_bytes: bad argument");
    const size_t pix = (size_t)B * H * Wd;
    *out_bytes = align_up(pix * kDims[stage] * 4) + align_up(pix * 4 * kDims[stage] * 4) + align_up(pix * 8);
    return ACX_OK;
}

int acx_block(acx_ctx* c, int stage, int block, float* x, int B, int H, int Wd, void* scratch, size_t scratch_bytes, void* stream) {
    ACX_TRY(need_ready(c));
    ACX_TRY(check_stage(stage, block));
    if (!x || !scratch) ACX_FAIL(ACX_ERR_ARG, "acx_block: null pointer");
    if (Wd != (kStemW >> stage)) ACX_FAIL(ACX_ERR_SHAPE, "stage %d has width %d, got %d", stage, kStemW >> stage, Wd);
    size_t need = 0;
    ACX_TRY(acx_block_scratch_bytes(stage, B, H, Wd, &need));
    if (scratch_bytes < need || ((uintptr_t)scratch & 255)) ACX_FAIL(ACX_ERR_WORKSPACE, "acx_block: scratch too small (%zu < %zu) or misaligned", scratch_bytes, need);
    const size_t pix = (size_t)B * H * Wd;
    char* ws = (char*)scratch;
    float* y = (float*)ws;
    float* hidden = (float*)(ws + align_up(pix * kDims[stage] * 4));
    float* stats = (float*)(ws + align_up(pix * kDims[stage] * 4) + align_up(pix * 4 * kDims[stage] * 4));
    if (act_bf16(c, stage)) {
        // The ABI keeps fp32 tensors; the block itself runs on bf16 activations: x is rounded to bf16 into the (otherwise idle)
        // hidden scratch, dwconv + fused MLP read and write bf16 there, and the result is widened back into x.
        const long long n = (long long)pix * kDims[stage];
        ACX_TRY(launch_convert_f32_to_bf16(x, hidden, n, (hipStream_t)stream));
        ACX_TRY(run_block(c, stage, block, hidden, y, hidden, stats, B, H, Wd, (hipStream_t)stream));
        return launch_convert_bf16_to_f32(hidden, x, n, (hipStream_t)stream);
    }
    return run_block(c, stage, block, x, y, hidden, stats, B, H, Wd, (hipStream_t)stream);
}

int acx_downsample(acx_ctx* c, int i, const float* x, float* out, float* scratch, int B, int H, int Wd, void* stream) {
    ACX_TRY(need_ready(c));
    if (i < 1 || i > 3) ACX_FAIL(ACX_ERR_ARG, "acx_downsample: layer index %d (expected 1..3)", i);
    if (!x || !out || !scratch || B <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_downsample: bad argument");
    if (H < 2 || Wd < 2) ACX_FAIL(ACX_ERR_SHAPE, "acx_downsample: kernel size can't be greater than actual input size (%dx%d)", H, Wd);
    return run_downsample(c, i, x, out, scratch, B, H, Wd, (hipStream_t)stream);
}

int acx_pool_head(acx_ctx* c, const float* x, int B, int H3, float* scene, float* logits, float* probs, void* stream) {
    ACX_TRY(need_ready(c));
    if (!x || B <= 0 || H3 <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_pool_head: bad argument");
    return launch_pool_head(c, x, B, H3, scene, logits, probs, (hipStream_t)stream);
}

int acx_nhwc_to_nchw(const float* x, float* out, int B, int H, int Wd, int C, void* stream) {
    if (!x || !out || B <= 0 || H <= 0 || Wd <= 0 || C <= 0) ACX_FAIL(ACX_ERR_ARG, "acx_nhwc_to_nchw: bad argument");
    return launch_nhwc_to_nchw(nullptr, x, out, B, H, Wd, C, (hipStream_t)stream);
}

int acx_pcm16_to_f32(const int16_t* pcm, float* out, int64_t n, void* stream) {
    if (!pcm || !out || n < 0) ACX_FAIL(ACX_ERR_ARG, "acx_pcm16_to_f32: bad argument");
    if (n == 0) return ACX_OK;
    return launch_pcm16_to_f32(pcm, out, n, (hipStream_t)stream);
}

int acx_frontend_info(const acx_ctx* c, int* dense_dft, float* stft_deviation, int* mel_taps) {
    ACX_TRY(need_ready(c));
    if (dense_dft) *dense_dft = c->dense_stft ? 1 : 0;
    if (stft_deviation) *stft_deviation = c->stft_deviation;
    if (mel_taps) *mel_taps = c->mel_w_len;
    return ACX_OK;
}

int acx_tuning_refresh(void) {
    tuning_reload();
    return ACX_OK;
}

int acx_test_fail_sub(acx_ctx* c, int sub) {
    if (!c) ACX_FAIL(ACX_ERR_ARG, "null context");
    if (sub < -1 || sub >= acx_ctx::kMaxSplitWays) ACX_FAIL(ACX_ERR_ARG, "acx_test_fail_sub: sub-batch index %d", sub);
    c->fail_sub.store(sub, std::memory_order_relaxed);
    return ACX_OK;
}

int acx_profile_enable(acx_ctx* c, int on) {
    if (!c) ACX_FAIL(ACX_ERR_ARG, "null context");
    c->prof.on = on != 0;
    return ACX_OK;
}

int acx_profile_read(acx_ctx* c, double* ms, int64_t* launches) {
    if (!c || !ms || !launches) ACX_FAIL(ACX_ERR_ARG, "acx_profile_read: null argument");
    for (int k = 0; k < ACX_K_COUNT; ++k) { ms[k] = 0.0; launches[k] = 0; }
    for (auto& r : c->prof.recs) {
        if (r.a && r.b) {
            ACX_HIP(hipEventSynchronize(r.b));
            float t = 0.f;
            ACX_HIP(hipEventElapsedTime(&t, r.a, r.b));
            if (r.cls >= 0 && r.cls < ACX_K_COUNT) { ms[r.cls] += t; launches[r.cls] += 1; }
        }
        if (r.a) c->prof.pool.push_back(r.a);
        if (r.b) c->prof.pool.push_back(r.b);
    }
    c->prof.recs.clear();
    return ACX_OK;
}

}  // extern "C"
