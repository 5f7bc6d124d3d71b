// Shared host/device pieces of the 1024-point real FFT used by the log-mel kernel (K1).
// Kept free of HIP-only constructs so that tests/host_fft_check.cpp can run the very same
// index math and butterflies on the CPU (there is no GPU in the build container).
//
// Algorithm: 1024 real samples -> 512 complex z[n] = x[2n] + i x[2n+1] -> radix-8 Stockham
// autosort FFT in 3 passes (Ns = 1, 8, 64), 64 lanes x 8 points -> split into the 513-bin
// spectrum of the real signal.
#pragma once

#if defined(__HIPCC__)
#define ACX_HD __host__ __device__ __forceinline__
#else
#define ACX_HD inline
#endif

namespace acx {

struct cf {
    float x, y;
};

ACX_HD cf cf_make(float x, float y) { cf r; r.x = x; r.y = y; return r; }
ACX_HD cf cf_add(cf a, cf b) { return cf_make(a.x + b.x, a.y + b.y); }
ACX_HD cf cf_sub(cf a, cf b) { return cf_make(a.x - b.x, a.y - b.y); }
ACX_HD cf cf_mul(cf a, cf b) { return cf_make(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
ACX_HD cf cf_mul_negi(cf a) { return cf_make(a.y, -a.x); }   // a * (-i)

// in-order forward DFT of 4 points: X[r] = sum_q v[q] exp(-2 pi i q r / 4)
ACX_HD void fft4(cf& v0, cf& v1, cf& v2, cf& v3) {
    cf t0 = cf_add(v0, v2), t1 = cf_sub(v0, v2);
    cf t2 = cf_add(v1, v3), t3 = cf_mul_negi(cf_sub(v1, v3));
    v0 = cf_add(t0, t2);
    v2 = cf_sub(t0, t2);
    v1 = cf_add(t1, t3);
    v3 = cf_sub(t1, t3);
}

// in-order forward DFT of 8 points
ACX_HD void fft8(cf* v) {
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    fft4(e0, e1, e2, e3);
    fft4(o0, o1, o2, o3);
    const float s = 0.70710678118654752440f;
    cf w1 = cf_make((o1.x + o1.y) * s, (o1.y - o1.x) * s);      // o1 * (s, -s)
    cf w2 = cf_mul_negi(o2);                                      // o2 * (-i)
    cf w3 = cf_make((o3.y - o3.x) * s, -(o3.x + o3.y) * s);     // o3 * (-s, -s)
    v[0] = cf_add(e0, o0); v[4] = cf_sub(e0, o0);
    v[1] = cf_add(e1, w1); v[5] = cf_sub(e1, w1);
    v[2] = cf_add(e2, w2); v[6] = cf_sub(e2, w2);
    v[3] = cf_add(e3, w3); v[7] = cf_sub(e3, w3);
}

// One Stockham radix-8 pass of the 512-point FFT for lane j (0..63).
//   v[r] arrives holding in[j + 64 r]; on return v[r] belongs at out[dst + r * Ns].
//   tw1024[n] = exp(-2 pi i n / 1024); the 512-point twiddle exp(-2 pi i m / 512) = tw1024[2 m].
ACX_HD int fft512_pass(cf* v, int j, int Ns, const cf* tw1024) {
    const int k = j % Ns;
    if (Ns > 1) {
        const int step = k * (64 / Ns);     // twiddle for point r: exp(-2 pi i k r / (8 Ns)) = W512^(k r 64/Ns)
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cf_mul(v[r], tw1024[2 * step * r]);
    }
    fft8(v);
    return (j / Ns) * Ns * 8 + k;
}

// The same pass with the lane's seven twiddles already fetched: twr[r - 1] = tw1024[2 (j % Ns) (64 / Ns) r], r = 1..7
// (they depend on the lane and the pass only, not on the frame: the kernel keeps them in registers).
ACX_HD int fft512_pass_tw(cf* v, int j, int Ns, const cf* twr) {
#pragma unroll
    for (int r = 1; r < 8; ++r) v[r] = cf_mul(v[r], twr[r - 1]);
    fft8(v);
    return (j / Ns) * Ns * 8 + (j % Ns);
}

// Padded index of complex element i in the LDS exchange buffers: one pad slot every 8 elements.  Turns the
// 8- and 16-way bank conflicts of the radix-8 scatter (strides of 64 B and 512 B) into conflict-free
// (72-B / 576-B strides); the buffers hold 512 + 64 slots.
ACX_HD int fft_pad(int i) { return i + (i >> 3); }
constexpr int kFftBufSlots = 576;

// Bin k (0..512) of the 1024-point real FFT from the 512-point FFT Z of the packed signal (Z padded).
ACX_HD cf rfft1024_bin(const cf* Z, int k, const cf* tw1024) {
    cf a = Z[fft_pad(k & 511)];
    cf b = Z[fft_pad((512 - k) & 511)];
    b.y = -b.y;                                   // conj
    cf e = cf_make(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
    cf d = cf_make(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
    cf o = cf_mul_negi(d);
    return cf_add(e, cf_mul(tw1024[k], o));
}

// Bin k of the real FFT with the split twiddle tw1024[k] passed in (see rfft1024_bin)
ACX_HD cf rfft1024_bin_tw(const cf* Z, int k, cf twk) {
    cf a = Z[fft_pad(k & 511)];
    cf b = Z[fft_pad((512 - k) & 511)];
    b.y = -b.y;
    cf e = cf_make(0.5f * (a.x + b.x), 0.5f * (a.y + b.y));
    cf d = cf_make(0.5f * (a.x - b.x), 0.5f * (a.y - b.y));
    cf o = cf_mul_negi(d);
    return cf_add(e, cf_mul(twk, o));
}

// reflect padding without edge repeat (F.pad mode="reflect"): padded index p -> source index
ACX_HD long long reflect_index(long long p, long long L) {
    long long i = p - 512;
    if (i < 0) i = -i;
    if (i >= L) i = 2 * (L - 1) - i;
    return i;
}

}  // namespace acx
