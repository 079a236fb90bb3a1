// Scalar building blocks shared by every kernel of the hot path: complex128 arithmetic and
// the closed-form pieces of the reference's per-frequency arithmetic.
//
// Everything here is `HD` (host + device) so the same source is (a) inlined into the gfx950
// kernels and (b) instantiated on the host by tests/emul/ to unit-test kernel arithmetic in the
// GPU-less build container.  The product library never runs these on the host.
//
// Reference files restated (see each function): MTFwdSolver/mt1DField.jl:23-98,
// MTSensitivity/MT1DSensitivity.jl:25-243, MTFwdSolver/mt2DTE.jl:153-210, mt2DTM.jl:152-210,
// MTSensitivity/dataFuncSens.jl:21-123,197-298.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif

namespace hmcmt {

constexpr double MU0 = 4.0 * 3.14159265358979323846 * 1e-7;   // MT2DFwdSolver.jl:76
constexpr double EPS0 = 8.85 * 1e-12;                          // mt1DField.jl:35
constexpr double TWO_PI = 2.0 * 3.14159265358979323846;

struct alignas(16) cplx {
    double re, im;
};

HD cplx C(double re, double im = 0.0) { return cplx{re, im}; }
HD cplx operator+(cplx a, cplx b) { return cplx{a.re + b.re, a.im + b.im}; }
HD cplx operator-(cplx a, cplx b) { return cplx{a.re - b.re, a.im - b.im}; }
HD cplx operator-(cplx a) { return cplx{-a.re, -a.im}; }
HD cplx operator*(cplx a, cplx b) { return cplx{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
HD cplx operator*(double a, cplx b) { return cplx{a * b.re, a * b.im}; }
HD cplx operator*(cplx b, double a) { return cplx{a * b.re, a * b.im}; }
HD cplx& operator+=(cplx& a, cplx b) { a.re += b.re; a.im += b.im; return a; }
HD cplx& operator-=(cplx& a, cplx b) { a.re -= b.re; a.im -= b.im; return a; }
HD cplx conj(cplx a) { return cplx{a.re, -a.im}; }
HD cplx mul_i(cplx a) { return cplx{-a.im, a.re}; }           // i*a
HD double cabs2(cplx a) { return a.re * a.re + a.im * a.im; }
HD double cabs_(cplx a) { return hypot(a.re, a.im); }
HD bool cisnan(cplx a) { return isnan(a.re) || isnan(a.im); }

// Smith's division (robust against intermediate overflow of |b|^2).
HD cplx operator/(cplx a, cplx b) {
    if (fabs(b.re) >= fabs(b.im)) {
        double r = b.im / b.re, d = b.re + b.im * r;
        return cplx{(a.re + a.im * r) / d, (a.im - a.re * r) / d};
    }
    double r = b.re / b.im, d = b.re * r + b.im;
    return cplx{(a.re * r + a.im) / d, (a.im * r - a.re) / d};
}
HD cplx operator/(cplx a, double b) { return cplx{a.re / b, a.im / b}; }
HD cplx operator/(double a, cplx b) { return cplx{a, 0.0} / b; }
HD cplx crecip(cplx b) { return cplx{1.0, 0.0} / b; }
// 1/b = conj(b)/|b|^2 for operands whose square neither overflows nor underflows (the FDM pivots: |b| within 1e-100 .. 1e100 by
// construction): ONE division on the dependent chain where Smith's form has two and a data-dependent branch -- the serial pivot
// recurrence (k_pivot) is a chain of these
HD cplx crecip_plain(cplx b) { const double i2 = 1.0 / (b.re * b.re + b.im * b.im); return cplx{b.re * i2, -(b.im * i2)}; }

// principal square root
HD cplx csqrt_(cplx z) {
    double m = hypot(z.re, z.im);
    if (m == 0.0) return cplx{0.0, 0.0};
    double t;
    if (z.re >= 0.0) {
        t = sqrt(0.5 * (m + z.re));
        return cplx{t, 0.5 * z.im / t};
    }
    t = sqrt(0.5 * (m - z.re));
    return cplx{0.5 * fabs(z.im) / t, z.im >= 0.0 ? t : -t};
}

HD cplx cexp_(cplx z) {
    double e = exp(z.re), s, c;
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(z.im, &s, &c);
#else
    s = sin(z.im); c = cos(z.im);
#endif
    return cplx{e * c, e * s};
}

// tanh(x+iy) = (sinh 2x + i sin 2y)/(cosh 2x + cos 2y), saturated for |x| large
HD cplx ctanh_(cplx z) {
    if (fabs(z.re) > 20.0) {
        // cosh(2x) ~ e^{2|x|}/2 dominates: tanh -> sign(x) + i*4 sin(2y) e^{-2|x|}/... (below 1e-17)
        double s = z.re > 0.0 ? 1.0 : -1.0;
        double e = exp(-2.0 * fabs(z.re));
        return cplx{s, 2.0 * sin(2.0 * z.im) * e};
    }
    double sh = sinh(2.0 * z.re), ch = cosh(2.0 * z.re);
    double d = ch + cos(2.0 * z.im);
    return cplx{sh / d, sin(2.0 * z.im) / d};
}

// ---------------------------------------------------------------------------------------------
// 1-D layered-earth fields used as Dirichlet values (mt1DField.jl:23-98).
//
// The transcendental work (one complex sqrt, tanh and two exps per layer) is independent across
// layers, so it is done by a massively parallel table kernel (layer_forward); the two serial
// recurrences then only multiply and divide.  k^2 keeps the displacement term mu0*eps0*omega^2
// (mt1DField.jl:48,52,66).
// ---------------------------------------------------------------------------------------------
HD cplx k_eps(double sig, double omega) {
    return csqrt_(cplx{MU0 * EPS0 * omega * omega, -MU0 * sig * omega});
}

// Per-layer terms of one column, everything of the two recurrences that does not depend on the running state
// (so the serial kernel is left with two 2x2 complex matrix-vector products per layer):
//   out[0] k                                  out[1] zp = w mu0 / k            out[2] th = tanh(i k h)
//   out[3] zp * (zp * th)                     -- impedance recurrence (mt1DField.jl:48-56) in projective form
//   out[4..7] m11 m12 m21 m22 = (pInv*eUD)*e  -- amplitude propagation into the next layer (:69-75), with
//             kr = k / k_next (1 for the last layer: half-space copy), e = exp(+-i k h)
constexpr int FWD_NQ = 8;
// exp(+-ikh) and tanh(ikh) of one layer from ONE exponential and ONE sine/cosine pair (round 3; ctanh_ + two cexp_ were five
// exponentials and four sine/cosine evaluations -- the per-layer terms are 350 000 independent items per evaluation on the
// headline mesh and fp64-issue-bound).  With ikh = a + ib:  e+- = e^{+-a} (cos b +- i sin b)  and
//   tanh(a + ib) = (sinh a cosh a + i sin b cos b) / (sinh^2 a + cos^2 b),
// sinh a from expm1 (no cancellation in thin or resistive layers), saturated like ctanh_ for |a| > 20.
struct LayerExp { cplx ep, em, th; };
HD LayerExp layer_exp(cplx k, double h) {
    const double a = -k.im * h, b = k.re * h;
    double s, c;
#if defined(__HIP_DEVICE_COMPILE__)
    sincos(b, &s, &c);
#else
    s = sin(b); c = cos(b);
#endif
    LayerExp r;
    double E, Ei;
    if (fabs(a) > 20.0) {
        E = exp(a); Ei = exp(-a);
        const double t = a > 0.0 ? Ei : E;
        r.th = cplx{a > 0.0 ? 1.0 : -1.0, 4.0 * s * c * t * t};
    } else {
        const double em1 = expm1(a);
        E = em1 + 1.0; Ei = 1.0 / E;
        const double sh = 0.5 * (em1 + em1 * Ei), ch = 0.5 * (E + Ei);
        const double den = sh * sh + c * c;
        r.th = cplx{sh * ch / den, s * c / den};
    }
    r.ep = cplx{E * c, E * s};
    r.em = cplx{Ei * c, -(Ei * s)};
    return r;
}
// amplitude propagation into the next layer from k, k of the layer below and e+-
HD void layer_matrix(cplx k, cplx kNext, bool lastLayer, cplx ep, cplx em, cplx& m11, cplx& m12, cplx& m21, cplx& m22) {
    const cplx one = cplx{1.0, 0.0};
    const cplx kr = lastLayer ? one : k * crecip(kNext);
    m11 = (0.5 * (one + kr)) * ep; m12 = (0.5 * (one - kr)) * em;
    m21 = (0.5 * (one - kr)) * ep; m22 = (0.5 * (one + kr)) * em;
}
HD void layer_forward(double sig, double sigNext, bool lastLayer, double omega, double h, cplx out[FWD_NQ]) {
    const cplx k = k_eps(sig, omega);
    const cplx zp = (omega * MU0) * crecip(k);
    const LayerExp x = layer_exp(k, h);
    out[0] = k;
    out[1] = zp;
    out[2] = x.th;
    out[3] = zp * (zp * x.th);
    layer_matrix(k, lastLayer ? k : k_eps(sigNext, omega), lastLayer, x.ep, x.em, out[4], out[5], out[6], out[7]);
}

// The same terms in the two groups k_bc_fused builds them in (LDS, one group in the space of the other): first k, e+ and
// the impedance recurrence's zp, th, zp*(zp*th) ...
HD void layer_up_terms(double sig, double omega, double h, cplx& k, cplx& ep, cplx& zp, cplx& th, cplx& zt) {
    k = k_eps(sig, omega);
    zp = (omega * MU0) * crecip(k);
    const LayerExp x = layer_exp(k, h);
    ep = x.ep; th = x.th;
    zt = zp * (zp * th);
}
// ... then the amplitude propagation's m11, m12, m21, m22 from the stored k, k of the layer below and e+
// (e- = conj(e+) / |e+|^2; zero where e+ has overflowed, as exp(-a) is)
HD void layer_down_terms(cplx k, cplx kNext, bool lastLayer, cplx ep, cplx& m11, cplx& m12, cplx& m21, cplx& m22) {
    const double n2 = cabs2(ep);
    cplx em = cplx{0.0, 0.0};
    if (n2 > 0.0 && n2 < 1e300) { const double i2 = 1.0 / n2; em = cplx{ep.re * i2, -(ep.im * i2)}; }
    layer_matrix(k, kNext, lastLayer, ep, em, m11, m12, m21, m22);
}

// Serial part for one column.  T points at this column's k-entry of layer 0; the FWD_NQ quantities are
// `qs` elements apart and consecutive layers `ls` elements apart.  outf(i, value) receives F_{i+1}/F_0, the
// normalised field below layer i (E for TE, H for TM, mt2DTE.jl:115-124); returns F_nz/F_0.
// Table entries are fetched RB layers at a time before the dependent arithmetic of those layers starts.
// (On the device the edge lanes buffer the outf values in LDS: a global store per layer inside the loop would sit
// in the same in-order vmcnt queue as the table loads.)
constexpr int RB = 4;
constexpr int RBF = 8;             // layers per block of the forward recurrences (4: 92 us, 8: 68 us, 16: 75 us for k_bc_forward)
// Both polarisations of one frequency in one pass: the layered-earth recurrences (impedance bottom -> top, amplitudes
// top -> bottom) are the same for TE and TM; only the functional of the amplitudes differs -- E = Eu + Ed (TE) or
// H = (Ed - Eu) k / (omega mu0) (TM), each normalised by its own top value.  outE(i, v) / outH(i, v) receive the
// normalised field under layer i.
// The normalised outputs of a layer from its amplitudes (shared by the in-loop and the deferred evaluation)
struct FwdTop { cplx if0E, if0H; double iomu0; };
HD void fwd_outputs(const FwdTop& tp, cplx eu, cplx ed, cplx kj, bool dead, cplx& oE, cplx& oH) {
    cplx fnE = eu + ed, fnH = ((ed - eu) * kj) * tp.iomu0;
    if (dead) { fnE = cplx{0.0, 0.0}; fnH = cplx{0.0, 0.0}; }
    oE = fnE * tp.if0E; oH = fnH * tp.if0H;
}
// Impedance recurrence bottom -> top (:48-56) of one column; the tables of zp, th and zp*(zp*th) are `ls` elements
// apart from layer to layer.  Returns Z at the surface.
// Z_j = zp (Z + zp th) / (zp + Z th) is a Moebius map: it is carried projectively, Z = N/D, so the serial
// chain has no division (a robust complex division is ~3 dependent fp64 divides); N and D are rescaled
// by an exact power of two once per block, and divided once at the top.
HD cplx bc1d_up(int nz, const cplx* Tzp, const cplx* Tth, const cplx* Tzt, long ls) {
    const cplx one = cplx{1.0, 0.0};
    // half-space below the last layer with the last layer's conductivity
    cplx zn = Tzp[(long)(nz - 1) * ls], zd = one;
    auto rescale = [&]() {
        const int e = -ilogb(fmax(fmax(fabs(zd.re), fabs(zd.im)), fmax(fabs(zn.re), fabs(zn.im))));
        if (e > -1000 && e < 1000) {                     // (zero / inf / nan: leave alone, the division below reports it)
            zn = cplx{ldexp(zn.re, e), ldexp(zn.im, e)};
            zd = cplx{ldexp(zd.re, e), ldexp(zd.im, e)};
        }
    };
    int j0 = nz - 1;
    for (; j0 - (RBF - 1) >= 0; j0 -= RBF) {               // whole blocks, all entries requested up front
        cplx c1[RBF], c2[RBF], c3[RBF];
#pragma unroll
        for (int t = 0; t < RBF; ++t) {
            const long j = j0 - t;
            c1[t] = Tzp[j * ls]; c3[t] = Tth[j * ls]; c2[t] = Tzt[j * ls];
        }
#pragma unroll
        for (int t = 0; t < RBF; ++t) {
            const cplx nn = c1[t] * zn + c2[t] * zd;
            zd = c1[t] * zd + zn * c3[t];
            zn = nn;
        }
        rescale();
    }
    for (; j0 >= 0; --j0) {                              // the remaining layers
        const long j = j0;
        const cplx a1 = Tzp[j * ls], a3 = Tth[j * ls], a2 = Tzt[j * ls];
        const cplx nn = a1 * zn + a2 * zd;
        zd = a1 * zd + zn * a3;
        zn = nn;
    }
    rescale();
    return zn / zd;
}

// Amplitude propagation top -> bottom (:62-83) from the surface impedance ztmp.
// Core: amp(i, eu, ed, kj, dead) is called with the amplitudes under every layer i; the outputs of the LAST layer are
// returned.  A caller that needs every layer's outputs evaluates fwd_outputs in amp (bc1d_forward_tab_2) or stores the
// amplitudes and evaluates them afterwards, off the serial loop (the edge columns of k_bc_forward / k_bc_fused).
template <class Amp>
HD void bc1d_down(double omega, int nz, cplx ztmp, const cplx* Tk, const cplx* Tm11, const cplx* Tm12, const cplx* Tm21,
                  const cplx* Tm22, long ls, Amp amp, FwdTop& tp, cplx& lastE_, cplx& lastH_) {
    const double omu0 = omega * MU0;
    const cplx one = cplx{1.0, 0.0};
    // top-layer up/down-going amplitudes (:62-63)
    cplx kj = Tk[0];
    const cplx a = omu0 / (ztmp * kj);
    cplx eu = 0.5 * (one - a), ed = 0.5 * (one + a);
    const double iomu0 = 1.0 / omu0;
    tp.if0E = crecip(eu + ed); tp.if0H = crecip(((ed - eu) * kj) * iomu0); tp.iomu0 = iomu0;
    bool dead = false;
    // one layer i -> i+1 (:69-83)
    // The serial chain is the 2x2 product alone: the cut-off test (three more dependent levels) only feeds the `dead`
    // flag, which masks the outputs -- the amplitudes are advanced unconditionally (behind the cut-off their values
    // are never used), so the test runs beside the chain instead of on it.
    auto down = [&](int i, cplx kn, cplx a11, cplx a12, cplx a21, cplx a22) {
        const cplx nu = a11 * eu + a12 * ed;
        const cplx nd = a21 * eu + a22 * ed;
        const double e2 = cabs2(nu + nd), e1 = cabs2(eu + ed);       // |.|^2: same ordering as |.|
        dead = dead || e2 - e1 > 0.0 || isnan(e2);                   // overflow cut-off: zero from here down
        eu = nu; ed = nd; kj = kn;
        amp(i, eu, ed, kj, dead);
    };
    // whole blocks of RBF layers with all table entries requested up front and no conditions around them (inside a
    // conditional the compiler sinks the loads next to their use: a second memory round trip per block), then the
    // remaining layers one by one; the last layer of all takes kn = kj (half-space copy below it)
    int i0 = 0;
    for (; i0 + RBF < nz; i0 += RBF) {
        cplx kn_[RBF], m11[RBF], m12[RBF], m21[RBF], m22[RBF];
#pragma unroll
        for (int t = 0; t < RBF; ++t) {
            const long i = i0 + t;
            kn_[t] = Tk[(i + 1) * ls];
            m11[t] = Tm11[i * ls]; m12[t] = Tm12[i * ls]; m21[t] = Tm21[i * ls]; m22[t] = Tm22[i * ls];
        }
#pragma unroll
        for (int t = 0; t < RBF; ++t) down(i0 + t, kn_[t], m11[t], m12[t], m21[t], m22[t]);
    }
    for (; i0 < nz; ++i0) {
        const long i = i0;
        const cplx kn = (i0 + 1 >= nz) ? kj : Tk[(i + 1) * ls];
        down(i0, kn, Tm11[i * ls], Tm12[i * ls], Tm21[i * ls], Tm22[i * ls]);
    }
    fwd_outputs(tp, eu, ed, kj, dead, lastE_, lastH_);
}

// Both recurrences on one table T (this column's k-entry of layer 0; quantities qs apart, layers ls apart)
template <class Amp>
HD void bc1d_forward_core(double omega, int nz, const cplx* T, long qs, long ls, Amp amp, FwdTop& tp, cplx& lastE_, cplx& lastH_) {
    const cplx ztmp = bc1d_up(nz, T + qs, T + 2 * qs, T + 3 * qs, ls);
    bc1d_down(omega, nz, ztmp, T, T + 4 * qs, T + 5 * qs, T + 6 * qs, T + 7 * qs, ls, amp, tp, lastE_, lastH_);
}

// Both polarisations of one frequency, every layer's outputs through outE(i, v) / outH(i, v)
template <class OutE, class OutH>
HD void bc1d_forward_tab_2(double omega, int nz, const cplx* T, long qs, long ls, OutE outE, OutH outH, cplx& lastE_, cplx& lastH_) {
    FwdTop tp;
    bc1d_forward_core(omega, nz, T, qs, ls, [&](int i, cplx eu, cplx ed, cplx kj, bool dead) {
        cplx oE, oH;
        fwd_outputs(tp, eu, ed, kj, dead, oE, oH);
        outE(i, oE); outH(i, oH);
    }, tp, lastE_, lastH_);
}


// one polarisation (compH: the TM functional)
template <class OutF>
HD cplx bc1d_forward_tab_f(double omega, int nz, const cplx* T, long qs, long ls, bool compH, OutF outf) {
    cplx lE, lH;
    bc1d_forward_tab_2(omega, nz, T, qs, ls, [&](int i, cplx v) { if (!compH) outf(i, v); },
                       [&](int i, cplx v) { if (compH) outf(i, v); }, lE, lH);
    return compH ? lH : lE;
}

HD cplx bc1d_forward_tab(double omega, int nz, const cplx* T, long qs, long ls, bool compH, cplx* out, long ostride) {
    return bc1d_forward_tab_f(omega, nz, T, qs, ls, compH, [=](int i, cplx v) { if (out) out[(long)i * ostride] = v; });
}

// ---------------------------------------------------------------------------------------------
// 1-D boundary-field sensitivities (MT1DSensitivity.jl:25-243) in three stages:
//   layer_sens          per layer:   ka (no displacement term, :59), 1/ka, exp(+-i ka h), exp(-2 i ka h)
//   sens_profile        per profile: top impedance z1 and d z1/d sig[c] (compImpJacMatrix :188-243),
//                                    up/down-going amplitudes per row, mixing terms, cut-off row
//   bc1d_sens_column    per derivative column c: walks the rows of the reference's dense (nz+1) x nz
//                                    matrix dE (source 'E', TE) or dH (source 'H', TM)
// Derivative w.r.t. the appended half-space is dropped (:162-164); the overflow cut-off zeroes only
// the lower-right block (:145-155).
// ---------------------------------------------------------------------------------------------
HD cplx k_noeps(double sig, double omu) { return csqrt_(cplx{0.0, -omu * sig}); }

// out[0] ka, out[1] 1/ka, out[2] exp(i ka h), out[3] 1/exp(i ka h), out[4] exp(-2 i ka h)
HD void layer_sens(double sig, double omega, double h, cplx out[5]) {
    const cplx k = k_noeps(sig, omega * MU0);
    out[0] = k;
    out[1] = crecip(k);
    out[2] = cexp_(mul_i(k * h));
    out[3] = crecip(out[2]);
    out[4] = cexp_((cplx{0.0, -2.0} * k) * h);
}

// Arrays of one profile (unit stride): ka, kinv [nz+1] (entry nz = half-space copy), expt, expr, ex2 [nz],
// eu, ed [nz+1], mix [4][nz], dz1 [nz].  Returns the cut-off row (first row >= 1 whose amplitudes were
// rejected) or nz+1; *z1out = top impedance.  fout (nullable): field value F[row], row = 1..nz, written
// to fout[(row-1)*fstride] (fstride 0: only the bottom value survives).
HD int sens_profile(double omega, int nz, const double* zLen, bool srcH, const cplx* ka, const cplx* kinv,
                    const cplx* expt, const cplx* expr, const cplx* ex2, cplx* eu, cplx* ed, cplx* mix,
                    cplx* dz1, cplx* z1out, cplx* fout, long fstride) {
    const double omu = omega * MU0;
    const cplx one = cplx{1.0, 0.0}, iom = cplx{0.0, omu};
    const int nL = nz + 1;
    // --- compImpJacMatrix; scratch: eu <- dZ/dZ_below, ed <- dZ/dsigma per layer
    cplx Z = omu * kinv[nL - 1];
    for (int j0 = nL - 2; j0 >= 0; j0 -= RB) {
        cplx ki[RB], e2_[RB];
#pragma unroll
        for (int t = 0; t < RB; ++t) { const int j = j0 - t >= 0 ? j0 - t : 0; ki[t] = kinv[j]; e2_[t] = ex2[j]; }
#pragma unroll
        for (int t = 0; t < RB; ++t) {
            const int j = j0 - t;
            if (j >= 0) {
                const cplx Zt = omu * ki[t];
                const cplx dZt = (cplx{0.0, 0.5 * omu * omu}) * (ki[t] * ki[t] * ki[t]);
                const cplx iS = crecip(Zt + Z);
                const cplx ex = e2_[t];
                const cplx L = ((Zt - Z) * iS) * ex;
                const cplx i1L = crecip(one + L);
                const cplx dL = ((2.0 * Z) * (iS * iS)) * ex * dZt + ((cplx{0.0, -2.0 * zLen[j]}) * L) * ((-(iom) / 2.0) * ki[t]);
                eu[j] = ((4.0 * (Zt * Zt)) * ex) * ((i1L * iS) * (i1L * iS));
                ed[j] = dZt * (one - L) * i1L + (Zt * (-2.0)) * (i1L * i1L) * dL;
                Z = Zt * (one - L) * i1L;
            }
        }
    }
    const cplx z1 = Z;
    *z1out = z1;
    cplx prod = one;
    for (int c = 0; c < nz; ++c) { dz1[c] = prod * ed[c]; prod = prod * eu[c]; }
    // --- amplitudes (:63-92, :126-157)
    cplx u, d;
    if (!srcH) {
        const cplx a = omu / (z1 * ka[0]);
        u = 0.5 * (one - a); d = 0.5 * (one + a);
    } else {
        const cplx hu = 0.5 * (one - z1 * ka[0] / omu), hd = 0.5 * (one + z1 * ka[0] / omu);
        u = -(omu * kinv[0]) * hu; d = (omu * kinv[0]) * hd;
    }
    eu[0] = u; ed[0] = d;
    int dead = nz + 1;
    for (int j0 = 0; j0 < nz; j0 += RB) {
        cplx ka_[RB], kan_[RB], kin_[RB], et_[RB], er_[RB];
#pragma unroll
        for (int t = 0; t < RB; ++t) {
            const int j = j0 + t < nz ? j0 + t : nz - 1;
            ka_[t] = ka[j]; kan_[t] = ka[j + 1]; kin_[t] = kinv[j + 1]; et_[t] = expt[j]; er_[t] = expr[j];
        }
#pragma unroll
        for (int t = 0; t < RB; ++t) {
            const int j = j0 + t;
            if (j < nz) {
                cplx fval = cplx{0.0, 0.0};
                if (dead > nz) {
                    const cplx kr = (j + 1 < nz) ? ka_[t] * kin_[t] : one;   // half-space copy: exactly 1
                    const cplx m11 = (one + kr) * et_[t], m12 = (one - kr) * er_[t];
                    const cplx m21 = (one - kr) * et_[t], m22 = (one + kr) * er_[t];
                    mix[j] = m11; mix[nz + j] = m12; mix[2 * nz + j] = m21; mix[3 * nz + j] = m22;
                    const cplx nu = ((0.5 * (one + kr)) * et_[t]) * u + ((0.5 * (one - kr)) * er_[t]) * d;
                    const cplx nd = ((0.5 * (one - kr)) * et_[t]) * u + ((0.5 * (one + kr)) * er_[t]) * d;
                    eu[j + 1] = nu; ed[j + 1] = nd;      // kept un-zeroed: the derivative row j+1 uses them
                    const double e2 = cabs2(nu + nd), e1 = cabs2(u + d);
                    if (e2 - e1 > 0.0 || isnan(e2)) dead = j + 1;
                    else {
                        u = nu; d = nd;
                        fval = srcH ? ((nd - nu) * kan_[t]) / omu : (nu + nd);
                    }
                } else {
                    mix[j] = mix[nz + j] = mix[2 * nz + j] = mix[3 * nz + j] = cplx{0, 0};
                    eu[j + 1] = ed[j + 1] = cplx{0, 0};
                }
                if (fout) fout[(long)j * fstride] = fval;
            }
        }
    }
    return dead;
}

// Returns sum_{row=1..nz} dF[row][c] * w[(row-1)*wstride]  (w == nullptr: dF[nz][c], the bottom-row entry
// used for the mean profile).
// dFout (optional): the column's entries dF(row j+1, c), j = 0..nz-1, are also stored at dFout[j * dstride] (zero in
// the rows beyond the cut-off) -- the w-independent part, precomputed off the critical path (item_bcsens_pre).
HD cplx bc1d_sens_column(double omega, int nz, const double* zLen, bool srcH, int c, const cplx* ka,
                         const cplx* kinv, const cplx* expt, const cplx* expr, const cplx* eu, const cplx* ed,
                         const cplx* mix, const cplx* dz1, cplx z1, int dead, const cplx* w, long wstride,
                         cplx* dFout = nullptr, long dstride = 0) {
    const double omu = omega * MU0;
    const cplx one = cplx{1.0, 0.0};
    const cplx dkc = (cplx{0.0, -omu / 2.0}) * kinv[c];   // dka[c][c] = (-i omu/2)/ka[c]
    // --- top row (:63-92)
    const cplx dk0 = (c == 0) ? dkc : cplx{0.0, 0.0};
    cplx dEu, dEd;
    if (!srcH) {
        const cplx a = omu / (z1 * ka[0]);
        dEu = (0.5 * a) * (dz1[c] / z1 + dk0 * kinv[0]);
        dEd = -dEu;
    } else {
        dEu = 0.5 * (dz1[c] + (omu * (kinv[0] * kinv[0])) * dk0);
        dEd = 0.5 * (dz1[c] - (omu * (kinv[0] * kinv[0])) * dk0);
    }
    cplx acc = cplx{0.0, 0.0};
    const int last = dead <= nz ? dead : nz;              // rows beyond the cut-off row are zero
    // d(mix)/d sig[c] is non-zero in rows j = c-1 and j = c only.  Those two contributions depend on the profile
    // tables, not on the running derivative, so each lane evaluates them ONCE here instead of inside the row loop
    // (where, with one column per lane, some lane of the wave would take that branch in almost every row and the
    // whole wave would pay for it).
    cplx hU[2] = {cplx{0.0, 0.0}, cplx{0.0, 0.0}}, hD[2] = {cplx{0.0, 0.0}, cplx{0.0, 0.0}};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = c - 1 + q;                          // q = 0: c == j + 1, q = 1: c == j
        if (j >= 0 && j < last) {
            const cplx dkj = (q == 1) ? dkc : cplx{0.0, 0.0};
            const cplx dkn = (q == 0) ? dkc : cplx{0.0, 0.0};   // c <= nz-1, so never the half-space copy
            const cplx kr = (j + 1 < nz) ? ka[j] * kinv[j + 1] : one;
            const cplx dexpt = (q == 1) ? (mul_i(zLen[j] * expt[j])) * dkj : cplx{0.0, 0.0};
            const cplx dexpr = (q == 1) ? (-(mul_i(zLen[j] * expr[j]))) * dkj : cplx{0.0, 0.0};
            const cplx dkr = dkj * kinv[j + 1] - (ka[j] * (kinv[j + 1] * kinv[j + 1])) * dkn;
            const cplx dmix11 = (one + kr) * dexpt + expt[j] * dkr, dmix12 = (one - kr) * dexpr - expr[j] * dkr;
            const cplx dmix21 = (one - kr) * dexpt - expt[j] * dkr, dmix22 = (one + kr) * dexpr + expr[j] * dkr;
            hU[q] = 0.5 * (dmix11 * eu[j] + dmix12 * ed[j]);
            hD[q] = 0.5 * (dmix21 * eu[j] + dmix22 * ed[j]);
        }
    }
    for (int j0 = 0; j0 < last; j0 += RB) {               // row j -> j+1 (:126-157)
        cplx mx[RB][4], eun[RB], edn[RB], kan[RB], wv[RB];
#pragma unroll
        for (int t = 0; t < RB; ++t) {
            const int j = j0 + t < last ? j0 + t : last - 1;
            mx[t][0] = mix[j]; mx[t][1] = mix[nz + j]; mx[t][2] = mix[2 * nz + j]; mx[t][3] = mix[3 * nz + j];
            eun[t] = eu[j + 1]; edn[t] = ed[j + 1]; kan[t] = ka[j + 1];
            wv[t] = w ? w[(long)j * wstride] : cplx{0.0, 0.0};
        }
#pragma unroll
        for (int t = 0; t < RB; ++t) {
            const int j = j0 + t;
            if (j < last) {
                const int row = j + 1;
                cplx nEu = 0.5 * (mx[t][0] * dEu + mx[t][1] * dEd);
                cplx nEd = 0.5 * (mx[t][2] * dEu + mx[t][3] * dEd);
                const cplx dkn = (c == j + 1) ? dkc : cplx{0.0, 0.0};
                if (c == j + 1) { nEu += hU[0]; nEd += hD[0]; }
                else if (c == j) { nEu += hU[1]; nEd += hD[1]; }
                cplx dF;
                if (!srcH) dF = nEu + nEd;
                else dF = ((edn[t] - eun[t]) / omu) * dkn + (kan[t] / omu) * (nEd - nEu);
                if (row == dead && c > j) dF = cplx{0.0, 0.0};   // cut-off row keeps columns c <= j only
                dEu = nEu; dEd = nEd;
                if (dFout) dFout[(long)j * dstride] = dF;
                if (w) acc += dF * wv[t];
                else if (row == nz) acc = dF;
            }
        }
    }
    if (dFout) for (int j = last; j < nz; ++j) dFout[(long)j * dstride] = cplx{0.0, 0.0};
    return acc;
}

// ---------------------------------------------------------------------------------------------
// Receiver-layer functional and its exact derivative.
// F0/F1: the two node rows (iy = 0..ny) at the receiver depth; dy, sig1: widths / conductivities
// of the receiver-layer cells; dz1 its thickness.
// ---------------------------------------------------------------------------------------------
struct RxAux { int k; };

// TE: Hy0 at node k (1 <= k <= ny-1)  (mt2DTE.jl:169-193)
HD cplx te_Hy0(int k, double omega, const cplx* F0, const cplx* F1, const double* dy,
               const double* sig1, double dz1) {
    const cplx iw = cplx{0.0, omega};
    auto HzQ = [&](int c) -> cplx {
        cplx b0 = (F0[c + 1] - F0[c]) / dy[c] / iw;
        cplx b1 = (F1[c + 1] - F1[c]) / dy[c] / iw;
        return (0.75 * b0 + 0.25 * b1) / MU0;
    };
    cplx HyH = -((F1[k] - F0[k]) / dz1 / (iw * MU0));
    cplx ExQ = 0.75 * F0[k] + 0.25 * F1[k];
    double avl = 0.5 * dy[k - 1] + 0.5 * dy[k];
    double sv = (0.5 * (sig1[k - 1] * dy[k - 1]) + 0.5 * (sig1[k] * dy[k])) / avl;
    cplx dHzQ = (HzQ(k) - HzQ(k - 1)) / avl;
    return HyH - (dHzQ - sv * ExQ) * (0.5 * dz1);
}

// TM: Ey0 at node k (1 <= k <= ny-1)  (mt2DTM.jl:168-193)
HD cplx tm_Ey0(int k, double omega, const cplx* F0, const cplx* F1, const double* dy,
               const double* sig1, double dz1) {
    auto EzQ = [&](int c) -> cplx {
        cplx j0 = -((F0[c + 1] - F0[c]) / dy[c]);
        cplx j1 = -((F1[c + 1] - F1[c]) / dy[c]);
        return (0.75 * j0 + 0.25 * j1) / sig1[c];
    };
    cplx JyH = (F1[k] - F0[k]) / dz1;
    double avl = 0.5 * dy[k - 1] + 0.5 * dy[k];
    double rv = (0.5 * (dy[k - 1] / sig1[k - 1]) + 0.5 * (dy[k] / sig1[k])) / avl;
    cplx EyH = JyH * rv;
    cplx HxQ = 0.75 * F0[k] + 0.25 * F1[k];
    cplx dEzQ = (EzQ(k) - EzQ(k - 1)) / avl;
    return EyH - (dEzQ + (cplx{0.0, omega * MU0}) * HxQ) * (0.5 * dz1);
}

HD int clampk(int k, int ny) { return k < 1 ? 1 : (k > ny - 1 ? ny - 1 : k); }

// Forward impedance at one receiver.  (idn-1, idn) and the un-normalised weights dy2, dy1 come
// from `findfirst(x -> x > rxY, yNode)` (mt2DTE.jl:196-207).
HD cplx rx_impedance(bool tm, double omega, int ny, const cplx* F0, const cplx* F1, const double* dy,
                     const double* sig1, double dz1, int idn, double dy1, double dy2) {
    cplx aL = tm ? tm_Ey0(clampk(idn - 1, ny), omega, F0, F1, dy, sig1, dz1)
                 : te_Hy0(clampk(idn - 1, ny), omega, F0, F1, dy, sig1, dz1);
    cplx aR = tm ? tm_Ey0(clampk(idn, ny), omega, F0, F1, dy, sig1, dz1)
                 : te_Hy0(clampk(idn, ny), omega, F0, F1, dy, sig1, dz1);
    cplx aux = aL * dy2 + aR * dy1;                       // Hy (TE) / Ey (TM)
    cplx fld = F0[idn - 1] * dy2 + F0[idn] * dy1;         // Ex (TE) / Hx (TM)
    return tm ? aux / fld : fld / aux;
}

// arr[slot] += v for a run-time slot, as predicated adds over the (unrolled) slots: a dynamically indexed local
// array would live in scratch memory on the GPU (a memory round trip per access; k_rx: 21 us); slots outside
// [0, N) are ignored
#if defined(__HIP_DEVICE_COMPILE__)
#define HMCMT_UNROLL _Pragma("unroll")
#else
#define HMCMT_UNROLL
#endif
template <int N>
HD void slot_add(cplx (&arr)[N], int slot, cplx v) {
    // (every slot gets an addend, v or 0: written as `if (j == slot) arr[j] += v` the compiler folds the unrolled
    // chain back into one dynamically indexed access)
    HMCMT_UNROLL
    for (int j = 0; j < N; ++j) {
        const bool hit = j == slot;
        arr[j] += cplx{hit ? v.re : 0.0, hit ? v.im : 0.0};
    }
}

// Derivative of the impedance of one receiver w.r.t. the two node rows and the receiver-layer
// conductivities (dataFuncSens.jl:44-123 TE, :219-298 TM; Impedance branch :118-123, :293-298).
// (kL, kR, wL, wR) from `linearInterp` (sensUtils.jl:133-161, normalised weights).
// Outputs: node window n0 = min(clampk(kL),clampk(kR))-1 ... n0+3, coefficient arrays d0[4], d1[4]
// (dZ/dF0[n0+i], dZ/dF1[n0+i]); cell window c0 = n0 ... c0+2, dq[3] = dZ/dsig1[c0+i].
HD void rx_impedance_deriv(bool tm, double omega, int ny, const cplx* F0, const cplx* F1,
                           const double* dy, const double* sig1, double dz1,
                           int kL, int kR, double wL, double wR,
                           int* n0_out, cplx d0[4], cplx d1[4], cplx dq[3]) {
    const int kk[2] = {clampk(kL, ny), clampk(kR, ny)};
    const double ww[2] = {wL, wR};
    const int n0 = (kk[0] < kk[1] ? kk[0] : kk[1]) - 1;
    *n0_out = n0;
    // field at the receiver with normalised weights
    cplx fld = wL * F0[kL] + wR * F0[kR];
    cplx aux = cplx{0, 0};
    // d(aux)/dF and d(aux)/dsig accumulated in a0[], a1[], aq[]
    cplx a0[4], a1[4], aq[3];
    HMCMT_UNROLL
    for (int i = 0; i < 4; ++i) { a0[i] = cplx{0, 0}; a1[i] = cplx{0, 0}; }
    HMCMT_UNROLL
    for (int i = 0; i < 3; ++i) aq[i] = cplx{0, 0};
    HMCMT_UNROLL
    for (int t = 0; t < 2; ++t) {
        const int k = kk[t];
        const double w = ww[t];
        const int o = k - n0;                              // node k sits at window slot o (1 or 2)
        const double avl = 0.5 * dy[k - 1] + 0.5 * dy[k];
        const double hz = 0.5 * dz1;
        if (!tm) {
            aux += w * te_Hy0(k, omega, F0, F1, dy, sig1, dz1);
            const cplx iw = cplx{0.0, omega};
            const cplx g = 1.0 / (dz1 * (iw * MU0));       // dHyH/dF0[k] = +g, /dF1[k] = -g
            const double sv = (0.5 * (sig1[k - 1] * dy[k - 1]) + 0.5 * (sig1[k] * dy[k])) / avl;
            const cplx ak = 1.0 / (dy[k] * (iw * MU0)), akm = 1.0 / (dy[k - 1] * (iw * MU0));
            // Hy0 = HyH - (dHzQ - sv*ExQ)*hz
            slot_add<4>(a0, o, w * (g + (sv * hz) * cplx{0.75, 0} + (hz / avl) * 0.75 * (ak + akm)));
            slot_add<4>(a1, o, w * (-g + (sv * hz) * cplx{0.25, 0} + (hz / avl) * 0.25 * (ak + akm)));
            slot_add<4>(a0, o + 1, w * (-(hz / avl) * 0.75 * ak));
            slot_add<4>(a1, o + 1, w * (-(hz / avl) * 0.25 * ak));
            slot_add<4>(a0, o - 1, w * (-(hz / avl) * 0.75 * akm));
            slot_add<4>(a1, o - 1, w * (-(hz / avl) * 0.25 * akm));
            const cplx ExQ = 0.75 * F0[k] + 0.25 * F1[k];
            // d sv / d sig[k-1] = 0.5*dy[k-1]/avl ; cells k-1 -> slot o-1, k -> slot o
            slot_add<3>(aq, o - 1, w * (hz * (0.5 * dy[k - 1] / avl)) * ExQ);
            slot_add<3>(aq, o, w * (hz * (0.5 * dy[k] / avl)) * ExQ);
        } else {
            aux += w * tm_Ey0(k, omega, F0, F1, dy, sig1, dz1);
            const double rv = (0.5 * (dy[k - 1] / sig1[k - 1]) + 0.5 * (dy[k] / sig1[k])) / avl;
            const double bk = 1.0 / (dy[k] * sig1[k]), bkm = 1.0 / (dy[k - 1] * sig1[k - 1]);
            const cplx iwm = cplx{0.0, omega * MU0};
            // Ey0 = EyH - (dEzQ + iwm*HxQ)*hz ; EzQ_c = -(0.75 dF0_c + 0.25 dF1_c) * b_c
            slot_add<4>(a0, o, w * (cplx{-rv / dz1, 0} - (hz / avl) * 0.75 * (bk + bkm) * cplx{1, 0} - (hz * 0.75) * iwm));
            slot_add<4>(a1, o, w * (cplx{rv / dz1, 0} - (hz / avl) * 0.25 * (bk + bkm) * cplx{1, 0} - (hz * 0.25) * iwm));
            slot_add<4>(a0, o + 1, w * cplx{(hz / avl) * 0.75 * bk, 0});
            slot_add<4>(a1, o + 1, w * cplx{(hz / avl) * 0.25 * bk, 0});
            slot_add<4>(a0, o - 1, w * cplx{(hz / avl) * 0.75 * bkm, 0});
            slot_add<4>(a1, o - 1, w * cplx{(hz / avl) * 0.25 * bkm, 0});
            const cplx JyH = (F1[k] - F0[k]) / dz1;
            auto EzQ = [&](int c) -> cplx {
                cplx j0 = -((F0[c + 1] - F0[c]) / dy[c]);
                cplx j1 = -((F1[c + 1] - F1[c]) / dy[c]);
                return (0.75 * j0 + 0.25 * j1) / sig1[c];
            };
            // d rv/d sig[k-1] = -0.5*dy[k-1]/(sig^2*avl); d EzQ_c/d sig_c = -EzQ_c/sig_c
            slot_add<3>(aq, o - 1, w * (JyH * (-0.5 * dy[k - 1] / (sig1[k - 1] * sig1[k - 1] * avl)) -
                              (hz / avl) * (EzQ(k - 1) / sig1[k - 1])));
            slot_add<3>(aq, o, w * (JyH * (-0.5 * dy[k] / (sig1[k] * sig1[k] * avl)) +
                          (hz / avl) * (EzQ(k) / sig1[k])));
        }
    }
    // chain to Z.  TE: Z = fld/aux ; TM: Z = aux/fld.   d fld / dF0[kL] = wL, dF0[kR] = wR
    cplx f0c[4] = {cplx{0, 0}, cplx{0, 0}, cplx{0, 0}, cplx{0, 0}};
    slot_add<4>(f0c, kL - n0, cplx{wL, 0});
    slot_add<4>(f0c, kR - n0, cplx{wR, 0});
    if (!tm) {
        const cplx ia = 1.0 / aux, c2 = fld / (aux * aux);
        HMCMT_UNROLL
        for (int i = 0; i < 4; ++i) { d0[i] = ia * f0c[i] - c2 * a0[i]; d1[i] = -(c2 * a1[i]); }
        HMCMT_UNROLL
        for (int i = 0; i < 3; ++i) dq[i] = -(c2 * aq[i]);
    } else {
        const cplx ifl = 1.0 / fld, c2 = aux / (fld * fld);
        HMCMT_UNROLL
        for (int i = 0; i < 4; ++i) { d0[i] = ifl * a0[i] - c2 * f0c[i]; d1[i] = ifl * a1[i]; }
        HMCMT_UNROLL
        for (int i = 0; i < 3; ++i) dq[i] = ifl * aq[i];
    }
}

}  // namespace hmcmt
