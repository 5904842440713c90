// CPU emulation of the W=512 wavefront FFT (16 lanes x 16 points, one LDS transpose, real split)
// and of the radix-2 Stockham fallback, mirroring the index math of aps_amd/csrc/stft.hip with
// plain loops in place of lanes.  Checks both against a double precision DFT.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../aps_amd/csrc/fft_core.h"

using aps::cf;

static double check(const std::vector<cf>& got, const std::vector<double>& x, int W) {
  double worst = 0, scale = 0;
  for (int k = 0; k <= W / 2; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < W; ++n) {
      re += x[n] * std::cos(2 * M_PI * k * n / W);
      im -= x[n] * std::sin(2 * M_PI * k * n / W);
    }
    scale = std::fmax(scale, std::hypot(re, im));
    worst = std::fmax(worst, std::hypot(got[k].re - re, got[k].im - im));
  }
  return worst / scale;
}

int main() {
  const int W = 512, M = 256, PITCH = 17;
  std::vector<double> x(W);
  srand(7);
  for (auto& v : x) v = rand() / (double)RAND_MAX - 0.5;
  std::vector<cf> tw(256), sp(257), scr(16 * PITCH), Z(256), X(257);
  for (int m = 0; m < 256; ++m) tw[m] = {(float)std::cos(2 * M_PI * m / 256), (float)-std::sin(2 * M_PI * m / 256)};
  for (int k = 0; k <= 256; ++k) sp[k] = {(float)std::cos(2 * M_PI * k / 512), (float)-std::sin(2 * M_PI * k / 512)};
  // pass 1
  for (int j = 0; j < 16; ++j) {
    cf z[16];
    for (int n1 = 0; n1 < 16; ++n1) {
      int e0 = 2 * (16 * n1 + j);
      z[n1] = {(float)x[e0], (float)x[e0 + 1]};
    }
    aps::dft16<false>(z);
    for (int k1 = 0; k1 < 16; ++k1) scr[k1 * PITCH + j] = (k1 == 0) ? z[0] : aps::cmul(z[k1], tw[j * k1]);
  }
  // pass 2
  for (int j = 0; j < 16; ++j) {
    cf z[16];
    for (int n2 = 0; n2 < 16; ++n2) z[n2] = scr[j * PITCH + n2];
    aps::dft16<false>(z);
    for (int k2 = 0; k2 < 16; ++k2) Z[j + 16 * k2] = z[k2];
  }
  for (int k = 0; k < 256; ++k) X[k] = aps::r2c_split(Z[k], Z[(256 - k) & 255], sp[k]);
  X[256] = aps::r2c_split(Z[0], Z[0], sp[256]);
  double e1 = check(X, x, W);
  printf("wave fft512 rel err %.3e\n", e1);

  // merge (inverse split) round trip: Z' from X must equal Z
  double e3 = 0;
  for (int k = 0; k < 256; ++k) {
    cf wpos = {sp[k].re, -sp[k].im};
    cf z = aps::c2r_merge(X[k], X[256 - k], wpos);
    e3 = std::fmax(e3, std::hypot(z.re - Z[k].re, z.im - Z[k].im));
  }
  printf("merge round trip abs err %.3e\n", e3);

  // Stockham radix-2, W = 128
  const int W2 = 128;
  std::vector<double> y(W2);
  for (auto& v : y) v = rand() / (double)RAND_MAX - 0.5;
  std::vector<cf> b0(W2), b1(W2), t2(W2);
  for (int e = 0; e < W2; ++e) {
    t2[e] = {(float)std::cos(2 * M_PI * e / W2), (float)-std::sin(2 * M_PI * e / W2)};
    b0[e] = {(float)y[e], 0.f};
  }
  cf *src = b0.data(), *dst = b1.data();
  const int half = W2 / 2;
  for (int ns = 1; ns < W2; ns <<= 1) {
    const int tstep = half / ns;
    for (int i = 0; i < half; ++i) {
      const int k = i & (ns - 1);
      const int jj = ((i - k) << 1) + k;
      cf u0 = src[i], u1 = aps::cmul(src[i + half], t2[k * tstep]);
      dst[jj] = u0 + u1;
      dst[jj + ns] = u0 - u1;
    }
    std::swap(src, dst);
  }
  std::vector<cf> out(src, src + W2);
  double e2 = check(out, y, W2);
  printf("stockham128 rel err %.3e\n", e2);
  return (e1 < 1e-6 && e2 < 1e-6 && e3 < 1e-5) ? 0 : 1;
}
