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
// The column profile is sigma_j = wa*sa[j*stride] + wb*sb[j*stride] (wb = 0 for an edge column;
// wa,wb = width weights of the two adjacent columns for a bottom node, mt2DTE.jl:127-131).
// `out` (nullable, stride `ostride`) receives F_j/F_0 for j = 1..nz (E for TE, H for TM,
// mt2DTE.jl:115-124); the return value is F_nz/F_0 (bottom value).
// k^2 keeps the displacement term mu0*eps0*omega^2 (mt1DField.jl:48,52,66).
// ---------------------------------------------------------------------------------------------
HD cplx k_eps(double sig, double omega) {
    return csqrt_(cplx{MU0 * EPS0 * omega * omega, -MU0 * sig * omega});
}

HD cplx bc1d_forward(double omega, int nz, const double* zLen, const double* sa, const double* sb,
                     int stride, double wa, double wb, bool compH, cplx* out, int ostride) {
    const double omu0 = omega * MU0;
    auto sig_at = [&](int j) -> double {
        double s = wa * sa[(long)j * stride];
        if (wb != 0.0) s += wb * sb[(long)j * stride];
        return s;
    };
    // impedance recurrence bottom -> top (:48-56); half-space has the last layer's conductivity
    cplx k = k_eps(sig_at(nz - 1), omega);
    cplx ztmp = omu0 / k;
    for (int j = nz - 1; j >= 0; --j) {
        k = k_eps(sig_at(j), omega);
        cplx zp = omu0 / k;
        cplx th = ctanh_(mul_i(k * zLen[j]));
        ztmp = zp * (ztmp + zp * th) / (zp + ztmp * th);
    }
    // top-layer up/down-going amplitudes (:62-63); k is the top layer's wavenumber here
    cplx a = omu0 / (ztmp * k);
    cplx eu = 0.5 * (cplx{1.0, 0.0} - a);
    cplx ed = 0.5 * (cplx{1.0, 0.0} + a);
    cplx kj = k;
    cplx f0 = compH ? ((ed - eu) * kj) / omu0 : (eu + ed);
    cplx last = cplx{1.0, 0.0};
    bool dead = false;
    for (int i = 0; i < nz; ++i) {                      // :69-83, layer i -> i+1
        cplx fn = cplx{0.0, 0.0};
        if (!dead) {
            cplx kn = (i + 1 < nz) ? k_eps(sig_at(i + 1), omega) : kj;   // half-space copy
            cplx kr = kj / kn;
            cplx ikh = mul_i(kj * zLen[i]);
            cplx ep = cexp_(ikh), em = cexp_(-ikh);
            cplx one = cplx{1.0, 0.0};
            // (pInv*eUD)*e, same association as the reference
            cplx m11 = (0.5 * (one + kr)) * ep, m12 = (0.5 * (one - kr)) * em;
            cplx m21 = (0.5 * (one - kr)) * ep, m22 = (0.5 * (one + kr)) * em;
            cplx nu = m11 * eu + m12 * ed;
            cplx nd = m21 * eu + m22 * ed;
            double e2 = cabs_(nu + nd), e1 = cabs_(eu + ed);
            if (e2 - e1 > 0.0 || isnan(e2)) {
                dead = true;                             // overflow cut-off: zero from here down
            } else {
                eu = nu; ed = nd; kj = kn;
                fn = compH ? ((ed - eu) * kj) / omu0 : (eu + ed);
            }
        }
        last = fn / f0;
        if (out) out[(long)i * ostride] = last;
    }
    return last;
}

// ---------------------------------------------------------------------------------------------
// 1-D boundary-field sensitivities (MT1DSensitivity.jl:25-243), one derivative column per call.
//
// For the profile sig[j*stride] (j = 0..nz-1, half-space copy appended) this walks the rows of the
// reference's dense (nz+1) x nz matrix dE (source 'E', TE) or dH (source 'H', TM) for ONE column
// c = derivative w.r.t. sig[c], and returns  sum_{row=1..nz} dF[row][c] * w[(row-1)*wstride]
// (w == nullptr: returns dF[nz][c] alone, the bottom-row entry used for the mean profile).
// If `fout` is non-null it also receives the field values F[row], row = 1..nz (the `bc` the
// reference's getBCderivTM returns, compJacTMatVec.jl:309,315) -- no displacement term here
// (MT1DSensitivity.jl:59), derivative w.r.t. the appended half-space dropped (:162-164), overflow
// cut-off zeroes only the lower-right block (:145-155).
// ---------------------------------------------------------------------------------------------
HD cplx k_noeps(double sig, double omu) { return csqrt_(cplx{0.0, -omu * sig}); }

HD cplx bc1d_sens_column(double omega, int nz, const double* zLen, const double* sig, int stride,
                         bool srcH, int c, const cplx* w, int wstride, cplx* fout, int fstride) {
    const double omu = omega * MU0;
    const cplx one = cplx{1.0, 0.0};
    const int nL = nz + 1;                               // layers incl. the half-space copy
    auto sg = [&](int j) -> double { return sig[(long)(j < nz ? j : nz - 1) * stride]; };

    // --- compImpJacMatrix (:188-243): top impedance z1 and d z1 / d sig[c]
    cplx Z = cplx{0.0, 0.0}, prod = one, dsig_c = cplx{0.0, 0.0};
    const cplx iom = cplx{0.0, omu};
    for (int j = nL - 1; j >= 0; --j) {
        cplx k = csqrt_(-(iom * sg(j)));
        cplx Zt = omu / k;
        cplx dZt = (cplx{0.0, omu * omu}) / (2.0 * (k * k * k));
        if (j == nL - 1) {
            Z = Zt;                                       // derivative w.r.t. the copy is dropped
            continue;
        }
        cplx RI = (Zt - Z) / (Zt + Z);
        cplx ex = cexp_((cplx{0.0, -2.0} * k) * zLen[j]);
        cplx L = RI * ex;
        cplx Ztmp = Zt * (one - L) / (one + L);
        cplx dL = (2.0 * Z) / ((Zt + Z) * (Zt + Z)) * ex * dZt +
                  ((cplx{0.0, -2.0 * zLen[j]}) * L) * (-(iom) / 2.0 / k);
        cplx dZ_ZP1 = (4.0 * (Zt * Zt)) * ex / (((one + L) * (Zt + Z)) * ((one + L) * (Zt + Z)));
        cplx dZ_sig = dZt * (one - L) / (one + L) + (Zt * (-2.0)) / ((one + L) * (one + L)) * dL;
        if (j == c) dsig_c = dZ_sig;
        else if (j < c) prod = prod * dZ_ZP1;
        Z = Ztmp;
    }
    const cplx z1 = Z;
    const cplx dz1 = (c == 0) ? dsig_c : prod * dsig_c;   // zimpDeri[c] (:231-239)

    // --- top layer (:63-92)
    cplx ka = k_noeps(sg(0), omu);
    auto dka_of = [&](cplx kk) -> cplx { return (cplx{0.0, -omu / 2.0}) / kk; };
    cplx dk0 = (c == 0) ? dka_of(ka) : cplx{0.0, 0.0};    // dka[0][c]
    cplx eu, ed, dEu, dEd, dHu, dHd;
    if (!srcH) {
        cplx a = omu / (z1 * ka);
        eu = 0.5 * (one - a);
        ed = 0.5 * (one + a);
        dEu = (0.5 * a) * (dz1 / z1 + dk0 / ka);
        dEd = -dEu;
        dHu = -(eu / omu) * dk0 - (ka / omu) * dEu;
        dHd = (ed / omu) * dk0 + (ka / omu) * dEd;
    } else {
        cplx hu = 0.5 * (one - z1 * ka / omu);
        cplx hd = 0.5 * (one + z1 * ka / omu);
        eu = -(omu / ka) * hu;
        ed = (omu / ka) * hd;
        dHu = (-0.5 / omu) * (z1 * dk0 + ka * dz1);
        dHd = -dHu;
        dEu = 0.5 * (dz1 + (omu / (ka * ka)) * dk0);
        dEd = 0.5 * (dz1 - (omu / (ka * ka)) * dk0);
    }

    cplx acc = cplx{0.0, 0.0};
    bool dead = false;
    for (int j = 0; j < nz; ++j) {                        // row j -> j+1 (:126-157)
        const int row = j + 1;
        cplx dF = cplx{0.0, 0.0}, fval = cplx{0.0, 0.0};
        if (!dead) {
            cplx kn = k_noeps(sg(j + 1), omu);
            cplx dkj = (c == j) ? dka_of(ka) : cplx{0.0, 0.0};          // dka[j][c]
            cplx dkn = (c == j + 1) ? dka_of(kn) : cplx{0.0, 0.0};      // dka[j+1][c] (c<nz only)
            cplx expt = cexp_(mul_i(ka * zLen[j]));
            cplx expr = one / expt;
            cplx dexpt = (c == j) ? (mul_i(zLen[j] * expt)) * dkj : cplx{0.0, 0.0};
            cplx dexpr = (c == j) ? (-(mul_i(zLen[j] * expr))) * dkj : cplx{0.0, 0.0};
            cplx kr = ka / kn;
            cplx dkr = dkj / kn - (ka / (kn * kn)) * dkn;
            cplx mix11 = (one + kr) * expt, mix12 = (one - kr) * expr;
            cplx mix21 = (one - kr) * expt, mix22 = (one + kr) * expr;
            cplx dmix11 = (one + kr) * dexpt + expt * dkr;
            cplx dmix12 = (one - kr) * dexpr - expr * dkr;
            cplx dmix21 = (one - kr) * dexpt - expt * dkr;
            cplx dmix22 = (one + kr) * dexpr + expr * dkr;
            // eLayer[:, j+1] = (pInv*eUD)*eLayer[:, j]
            cplx nu = ((0.5 * (one + kr)) * expt) * eu + ((0.5 * (one - kr)) * expr) * ed;
            cplx nd = ((0.5 * (one - kr)) * expt) * eu + ((0.5 * (one + kr)) * expr) * ed;
            cplx nEu = 0.5 * (dmix11 * eu + mix11 * dEu + dmix12 * ed + mix12 * dEd);
            cplx nEd = 0.5 * (dmix21 * eu + mix21 * dEu + dmix22 * ed + mix22 * dEd);
            cplx nHu = -(nu / omu) * dkn - (kn / omu) * nEu;
            cplx nHd = (nd / omu) * dkn + (kn / omu) * nEd;
            double e2 = cabs_(nu + nd), e1 = cabs_(eu + ed);
            if (e2 - e1 > 0.0 || isnan(e2)) {
                dead = true;
                // rows > j+1 are never computed (zero); row j+1 keeps columns c <= j only
                if (c <= j) dF = srcH ? (nHu + nHd) : (nEu + nEd);
            } else {
                dF = srcH ? (nHu + nHd) : (nEu + nEd);
                fval = srcH ? ((nd - nu) * kn) / omu : (nu + nd);
            }
            eu = nu; ed = nd; dEu = nEu; dEd = nEd; dHu = nHu; dHd = nHd; ka = kn;
        }
        if (fout) fout[(long)j * fstride] = fval;
        if (w) acc += dF * w[(long)(row - 1) * wstride];
        else if (row == nz) acc = dF;
    }
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
    for (int i = 0; i < 4; ++i) { d0[i] = cplx{0, 0}; d1[i] = cplx{0, 0}; }
    for (int i = 0; i < 3; ++i) dq[i] = cplx{0, 0};
    // field at the receiver with normalised weights
    cplx fld = wL * F0[kL] + wR * F0[kR];
    cplx aux = cplx{0, 0};
    // d(aux)/dF and d(aux)/dsig accumulated in a0[], a1[], aq[]
    cplx a0[4], a1[4], aq[3];
    for (int i = 0; i < 4; ++i) { a0[i] = cplx{0, 0}; a1[i] = cplx{0, 0}; }
    for (int i = 0; i < 3; ++i) aq[i] = cplx{0, 0};
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
            a0[o] += w * (g + (sv * hz) * cplx{0.75, 0} + (hz / avl) * 0.75 * (ak + akm));
            a1[o] += w * (-g + (sv * hz) * cplx{0.25, 0} + (hz / avl) * 0.25 * (ak + akm));
            a0[o + 1] += w * (-(hz / avl) * 0.75 * ak);
            a1[o + 1] += w * (-(hz / avl) * 0.25 * ak);
            a0[o - 1] += w * (-(hz / avl) * 0.75 * akm);
            a1[o - 1] += w * (-(hz / avl) * 0.25 * akm);
            const cplx ExQ = 0.75 * F0[k] + 0.25 * F1[k];
            // d sv / d sig[k-1] = 0.5*dy[k-1]/avl ; cells k-1 -> slot o-1, k -> slot o
            aq[o - 1] += w * (hz * (0.5 * dy[k - 1] / avl)) * ExQ;
            aq[o] += w * (hz * (0.5 * dy[k] / avl)) * ExQ;
        } else {
            aux += w * tm_Ey0(k, omega, F0, F1, dy, sig1, dz1);
            const double rv = (0.5 * (dy[k - 1] / sig1[k - 1]) + 0.5 * (dy[k] / sig1[k])) / avl;
            const double bk = 1.0 / (dy[k] * sig1[k]), bkm = 1.0 / (dy[k - 1] * sig1[k - 1]);
            const cplx iwm = cplx{0.0, omega * MU0};
            // Ey0 = EyH - (dEzQ + iwm*HxQ)*hz ; EzQ_c = -(0.75 dF0_c + 0.25 dF1_c) * b_c
            a0[o] += w * (cplx{-rv / dz1, 0} - (hz / avl) * 0.75 * (bk + bkm) * cplx{1, 0} - (hz * 0.75) * iwm);
            a1[o] += w * (cplx{rv / dz1, 0} - (hz / avl) * 0.25 * (bk + bkm) * cplx{1, 0} - (hz * 0.25) * iwm);
            a0[o + 1] += w * cplx{(hz / avl) * 0.75 * bk, 0};
            a1[o + 1] += w * cplx{(hz / avl) * 0.25 * bk, 0};
            a0[o - 1] += w * cplx{(hz / avl) * 0.75 * bkm, 0};
            a1[o - 1] += w * cplx{(hz / avl) * 0.25 * bkm, 0};
            const cplx JyH = (F1[k] - F0[k]) / dz1;
            auto EzQ = [&](int c) -> cplx {
                cplx j0 = -((F0[c + 1] - F0[c]) / dy[c]);
                cplx j1 = -((F1[c + 1] - F1[c]) / dy[c]);
                return (0.75 * j0 + 0.25 * j1) / sig1[c];
            };
            // d rv/d sig[k-1] = -0.5*dy[k-1]/(sig^2*avl); d EzQ_c/d sig_c = -EzQ_c/sig_c
            aq[o - 1] += w * (JyH * (-0.5 * dy[k - 1] / (sig1[k - 1] * sig1[k - 1] * avl)) -
                              (hz / avl) * (EzQ(k - 1) / sig1[k - 1]));
            aq[o] += w * (JyH * (-0.5 * dy[k] / (sig1[k] * sig1[k] * avl)) +
                          (hz / avl) * (EzQ(k) / sig1[k]));
        }
    }
    // chain to Z.  TE: Z = fld/aux ; TM: Z = aux/fld.   d fld / dF0[kL] = wL, dF0[kR] = wR
    cplx f0c[4] = {cplx{0, 0}, cplx{0, 0}, cplx{0, 0}, cplx{0, 0}};
    if (kL - n0 >= 0 && kL - n0 < 4) f0c[kL - n0] += cplx{wL, 0};
    if (kR - n0 >= 0 && kR - n0 < 4) f0c[kR - n0] += cplx{wR, 0};
    if (!tm) {
        const cplx ia = 1.0 / aux, c2 = fld / (aux * aux);
        for (int i = 0; i < 4; ++i) { d0[i] = ia * f0c[i] - c2 * a0[i]; d1[i] = -(c2 * a1[i]); }
        for (int i = 0; i < 3; ++i) dq[i] = -(c2 * aq[i]);
    } else {
        const cplx ifl = 1.0 / fld, c2 = aux / (fld * fld);
        for (int i = 0; i < 4; ++i) { d0[i] = ifl * a0[i] - c2 * f0c[i]; d1[i] = ifl * a1[i]; }
        for (int i = 0; i < 3; ++i) dq[i] = ifl * aq[i];
    }
}

}  // namespace hmcmt
