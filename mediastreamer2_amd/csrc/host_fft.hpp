// host_fft.hpp -- host-side packed real FFT used at DESIGN time only
// (equalizer impulse responses).  Same algorithm and operation order as the
// reference's speex-flavoured kiss_fft float build (src/utils/kiss_fft.c
// butterflies :38-149, stage order :320-408, factorisation :412-435, twiddles
// :464-471; src/utils/kiss_fftr.c super-twiddles :68-81, packed inverse
// :261-296), written as an iterative decimation-in-time transform: one digit
// permutation, then radix-4/2 stages from the innermost factor outwards.
// Sizes on this path are powers of two (128/256/512).
#pragma once
#include <cmath>
#include <vector>

namespace mi {

struct Cpx {
	float r, i;
};

class ComplexFft {
  public:
	ComplexFft(int n, bool inverse) : n_(n), inverse_(inverse), tw_((size_t)n) {
		const double pi = 3.14159265358979323846264338327;
		for (int k = 0; k < n; ++k) {
			double phase = (-2 * pi / n) * k;
			if (inverse) phase *= -1;
			tw_[(size_t)k] = {(float)std::cos(phase), (float)std::sin(phase)};
		}
		int left = n;
		while (left > 1) { // 4s first, then a final 2
			const int p = (left % 4 == 0) ? 4 : 2;
			left /= p;
			radix_.push_back(p);
			rest_.push_back(left);
		}
		// digit permutation: out = sum j_L * rest_L, in = sum j_L * stride_L
		perm_.assign((size_t)n, 0);
		std::vector<int> stride(radix_.size());
		int f = 1;
		for (size_t L = 0; L < radix_.size(); ++L) {
			stride[L] = f;
			f *= radix_[L];
		}
		for (int o = 0; o < n; ++o) {
			int rem = o, src = 0;
			for (size_t L = 0; L < radix_.size(); ++L) {
				const int j = rem / rest_[L];
				rem -= j * rest_[L];
				src += j * stride[L];
			}
			perm_[(size_t)o] = src;
		}
		stride_ = stride;
	}

	void run(const Cpx *in, Cpx *out) const {
		for (int o = 0; o < n_; ++o) out[o] = in[perm_[(size_t)o]];
		for (int L = (int)radix_.size() - 1; L >= 0; --L) {
			const int p = radix_[(size_t)L], m = rest_[(size_t)L], fs = stride_[(size_t)L];
			const int span = p * m; // distance between the fs sub-transforms of this stage
			for (int b = 0; b < fs; ++b) {
				Cpx *F = out + (size_t)b * span;
				if (p == 2) stage2(F, m, fs);
				else stage4(F, m, fs);
			}
		}
	}

  private:
	static Cpx mul(const Cpx &a, const Cpx &b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }

	void stage2(Cpx *F, int m, int fs) const {
		for (int j = 0; j < m; ++j) {
			const Cpx t = mul(F[m + j], tw_[(size_t)j * fs]);
			F[m + j] = {F[j].r - t.r, F[j].i - t.i};
			F[j].r += t.r;
			F[j].i += t.i;
		}
	}

	void stage4(Cpx *F, int m, int fs) const {
		for (int j = 0; j < m; ++j) {
			const Cpx s0 = mul(F[m + j], tw_[(size_t)j * fs]);
			const Cpx s1 = mul(F[2 * m + j], tw_[(size_t)j * fs * 2]);
			const Cpx s2 = mul(F[3 * m + j], tw_[(size_t)j * fs * 3]);
			const Cpx s5 = {F[j].r - s1.r, F[j].i - s1.i};
			F[j].r += s1.r;
			F[j].i += s1.i;
			const Cpx s3 = {s0.r + s2.r, s0.i + s2.i};
			const Cpx s4 = {s0.r - s2.r, s0.i - s2.i};
			F[2 * m + j] = {F[j].r - s3.r, F[j].i - s3.i};
			F[j].r += s3.r;
			F[j].i += s3.i;
			if (inverse_) {
				F[m + j] = {s5.r - s4.i, s5.i + s4.r};
				F[3 * m + j] = {s5.r + s4.i, s5.i - s4.r};
			} else {
				F[m + j] = {s5.r + s4.i, s5.i - s4.r};
				F[3 * m + j] = {s5.r - s4.i, s5.i + s4.r};
			}
		}
	}

	int n_;
	bool inverse_;
	std::vector<Cpx> tw_;
	std::vector<int> radix_, rest_, stride_, perm_;
};

// packed spectrum [DC, Re1, Im1, ..., Nyquist] -> nfft real samples, unscaled (ms_ifft)
inline void packed_real_ifft(int nfft, const float *freq, float *time) {
	const int n = nfft / 2;
	ComplexFft sub(n, true);
	std::vector<Cpx> tmp((size_t)n), super((size_t)n), out((size_t)n);
	const double pi = 3.14159265358979323846264338327;
	for (int k = 0; k < n; ++k) {
		const double phase = pi * (((double)k) / n + .5);
		super[(size_t)k] = {(float)std::cos(phase), (float)std::sin(phase)};
	}
	tmp[0] = {freq[0] + freq[2 * n - 1], freq[0] - freq[2 * n - 1]};
	for (int k = 1; k <= n / 2; ++k) {
		const Cpx fk = {freq[2 * k - 1], freq[2 * k]};
		const Cpx fnkc = {freq[2 * (n - k) - 1], -freq[2 * (n - k)]};
		const Cpx fek = {fk.r + fnkc.r, fk.i + fnkc.i};
		const Cpx d = {fk.r - fnkc.r, fk.i - fnkc.i};
		const Cpx fok = {d.r * super[(size_t)k].r - d.i * super[(size_t)k].i,
		                 d.r * super[(size_t)k].i + d.i * super[(size_t)k].r};
		tmp[(size_t)k] = {fek.r + fok.r, fek.i + fok.i};
		Cpx c = {fek.r - fok.r, fek.i - fok.i};
		c.i *= -1;
		tmp[(size_t)(n - k)] = c;
	}
	sub.run(tmp.data(), out.data());
	for (int k = 0; k < n; ++k) {
		time[2 * k] = out[(size_t)k].r;
		time[2 * k + 1] = out[(size_t)k].i;
	}
}

} // namespace mi
