// CPU emulation of the log-mel kernel's FFT schedule (64 lanes x 3 radix-8 passes + real split),
// built from the same fft_core.h the HIP kernel uses.  Test infrastructure only.
#include <cmath>
#include <vector>
#include "../audioset-convnext-inf_amd/csrc/fft_core.h"
using namespace acx;

extern "C" void acx_host_power_spectrum(const float* xw /*1024 windowed samples*/, float* P /*513*/) {
    std::vector<cf> tw(1024);
    for (int n = 0; n < 1024; ++n) {
        double a = -2.0 * M_PI * n / 1024.0;
        tw[n] = cf_make((float)std::cos(a), (float)std::sin(a));
    }
    std::vector<cf> buf0(kFftBufSlots), buf1(kFftBufSlots);       // padded exactly as the kernel's LDS buffers
    for (int n = 0; n < 512; ++n) buf0[fft_pad(n)] = cf_make(xw[2 * n], xw[2 * n + 1]);
    cf* in = buf0.data();
    cf* out = buf1.data();
    for (int Ns = 1; Ns < 512; Ns *= 8) {
        for (int j = 0; j < 64; ++j) {
            cf v[8];
            for (int r = 0; r < 8; ++r) v[r] = in[fft_pad(j + 64 * r)];
            int dst = fft512_pass(v, j, Ns, tw.data());
            for (int r = 0; r < 8; ++r) out[fft_pad(dst + r * Ns)] = v[r];
        }
        cf* t = in; in = out; out = t;
    }
    for (int k = 0; k <= 512; ++k) {
        cf X = rfft1024_bin(in, k, tw.data());
        P[k] = X.x * X.x + X.y * X.y;
    }
}
extern "C" long long acx_host_reflect(long long p, long long L) { return reflect_index(p, L); }
