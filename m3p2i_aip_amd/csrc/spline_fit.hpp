// spline_fit.hpp -- smoothing-spline fit + evaluation of ONE data series, host and device.
//
// What it restates: P. Dierckx's FITPACK routine `curfit` (iopt = 0, unit weights) followed by
// `splev` (ext = 3), i.e. what the reference's sampler calls through scipy for every sample and
// control dimension (mppi_utils.py bspline(): si.splrep(x, y, k=degree, s=0.5) + si.splev).
// scipy (pinned in this image: 1.15.3) ships FITPACK only as a compiled library, so this is a
// restatement of the published algorithm (Dierckx, "Curve and Surface Fitting with Splines", 1993,
// ch. 4-5; routines fpcurf / fpknot / fpdisc / fpgivs / fprota / fpback / fprati / fpbspl), in the
// same operation order, validated against scipy itself (tests/test_spline_fit.py; on the GPU
// tests/test_device_sampler_gpu.py).  All arithmetic is binary64 like FITPACK's.
//
// Provenance: FITPACK ("DIERCKX") is third-party numerical software by Paul Dierckx, distributed through
// netlib (http://www.netlib.org/dierckx/) and shipped, compiled, inside SciPy under SciPy's BSD-3-Clause
// licence terms.  It is NOT part of tud-amr/m3p2i-aip (the reference only calls it through scipy), and no
// FITPACK source was available here (scipy ships the object code only): this file follows the algorithm as
// published, and its agreement with the compiled library is a test, not an assumption.
//
// One call = one series of m points y[0..m) at abscissae x_i = i * m / (m - 1)  (np.linspace(0, m, m),
// skill_utils/mppi_utils bspline), evaluated at n_out points linspace(0, m, n_out).
#pragma once

#ifdef __HIPCC__
#define M3_HD __host__ __device__
#else
#define M3_HD
#endif

namespace m3 {

constexpr int SF_MAX_M = 64;               // data points per series (horizon T <= 256)
constexpr int SF_MAX_K = 3;                // spline degree (the reference uses 2)
constexpr int SF_NEST = SF_MAX_M + SF_MAX_K + 1;
constexpr int SF_K2 = SF_MAX_K + 2;

// Givens rotation parameters (fpgivs)
M3_HD inline void sf_givs(double piv, double& ww, double& cs, double& sn) {
    const double store = piv < 0 ? -piv : piv;
    double dd;
    if (store >= ww) { const double r = ww / piv; dd = store * sqrt(1.0 + r * r); }
    else { const double r = piv / ww; dd = ww * sqrt(1.0 + r * r); }
    cs = ww / dd;
    sn = piv / dd;
    ww = dd;
}
// apply a Givens rotation (fprota)
M3_HD inline void sf_rota(double cs, double sn, double& a, double& b) {
    const double s1 = a, s2 = b;
    b = cs * s2 + sn * s1;
    a = cs * s1 - sn * s2;
}

// non-zero B-splines of degree k at x, t[l] <= x < t[l+1] (1-based l as in FITPACK) (fpbspl)
M3_HD inline void sf_bspl(const double* t /*1-based*/, int k, double x, int l, double* h /*1-based, k+1*/) {
    double hh[SF_MAX_K + 2];
    h[1] = 1.0;
    for (int j = 1; j <= k; ++j) {
        for (int i = 1; i <= j; ++i) hh[i] = h[i];
        h[1] = 0.0;
        for (int i = 1; i <= j; ++i) {
            const int li = l + i, lj = li - j;
            if (t[li] != t[lj]) {
                const double f = hh[i] / (t[li] - t[lj]);
                h[i] = h[i] + f * (t[li] - x);
                h[i + 1] = f * (x - t[lj]);
            } else {
                h[i + 1] = 0.0;
            }
        }
    }
}

// back substitution for an upper triangular band matrix a(n, k) (fpback)
M3_HD inline void sf_back(const double (*a)[SF_K2 + 1], const double* z, int n, int k, double* c) {
    const int k1 = k - 1;
    c[n] = z[n] / a[n][1];
    int i = n - 1;
    if (i == 0) return;
    for (int j = 2; j <= n; ++j) {
        double store = z[i];
        int i1 = k1;
        if (j <= k1) i1 = j - 1;
        int m = i;
        for (int l = 1; l <= i1; ++l) {
            m = m + 1;
            store = store - c[m] * a[i][l + 1];
        }
        c[i] = store / a[i][1];
        i = i - 1;
    }
}

// the root of the rational interpolant through (p1,f1), (p2,f2), (p3,f3); p3 <= 0 means infinity.
// Also moves the bracket (fprati).
M3_HD inline double sf_rati(double& p1, double& f1, double p2, double f2, double& p3, double& f3) {
    double p;
    if (p3 > 0.0) {
        const double h1 = f1 * (f2 - f3), h2 = f2 * (f3 - f1), h3 = f3 * (f1 - f2);
        p = -(p1 * p2 * h3 + p2 * p3 * h1 + p3 * p1 * h2) / (p1 * h1 + p2 * h2 + p3 * h3);
    } else {
        p = (p1 * (f1 - f3) * f2 - p2 * (f2 - f3) * f1) / ((f1 - f2) * f3);
    }
    if (f2 < 0.0) { p3 = p2; f3 = f2; }
    else { p1 = p2; f1 = f2; }
    return p;
}

// Fit (k, s) to y[0..m) on x = linspace(0, m, m) and evaluate at linspace(0, m, n_out) -> out[i*stride].
// Returns the number of knots n (>= 2k+2), or < 0 for invalid arguments.
template <typename OutT>
M3_HD inline int spline_fit_eval(const double* y0, int m, int k, double s, int n_out, OutT* out, int stride) {
    if (m <= k || m > SF_MAX_M || k < 1 || k > SF_MAX_K || n_out < 1) return -1;
    const double tol = 0.001, con1 = 0.1, con9 = 0.9, con4 = 0.04, half = 0.5;
    const int maxit = 20;
    const int k1 = k + 1, k2 = k1 + 1, nmin = 2 * k1;
    const int nest = (m + k + 1 > 2 * k + 3) ? m + k + 1 : 2 * k + 3;
    double x[SF_MAX_M + 1], y[SF_MAX_M + 1];
    for (int i = 1; i <= m; ++i) {
        x[i] = (m == 1) ? 0.0 : (double)(i - 1) * ((double)m / (double)(m - 1));   // np.linspace(0, m, m)
        y[i] = y0[i - 1];
    }
    x[m] = (double)m;
    const double xb = x[1], xe = x[m];
    double t[SF_NEST + 2], c[SF_NEST + 2], z[SF_NEST + 2], fpint[SF_NEST + 2];
    int nrdata[SF_NEST + 2];
    double a[SF_NEST + 1][SF_K2 + 1], g[SF_NEST + 1][SF_K2 + 1], b[SF_NEST + 1][SF_K2 + 1];
    double q[SF_MAX_M + 1][SF_MAX_K + 2];
    double h[SF_K2 + 2];
    const double acc = tol * s;
    const int nmax = m + k1;
    int n = nmin, nk1 = n - k1, nrint = 0, nplus = 0;
    bool poly = false;       // FITPACK's ier == -2: current fit is the no-interior-knot polynomial
    bool first = true;       // FITPACK's "ier != 0" on the first pass of the knot loop
    bool interp = false;     // n reached nmax: interpolating spline (ier = -1)
    double fp = 0.0, fpold = 0.0, fp0 = 0.0, fpms = 0.0;
    nrdata[1] = m - 2;

    // ---- part 1: number and position of the knots ---------------------------------------
    bool place_interp_knots = false;
    for (int iter = 1; iter <= m; ++iter) {
        if (place_interp_knots) {
            // knots of the interpolating spline (n == nmax)
            const int mk1 = m - k1;
            if (mk1 != 0) {
                const int k3 = k / 2;
                int i = k2, j = k3 + 2;
                if (k3 * 2 == k) {
                    for (int l = 1; l <= mk1; ++l) { t[i] = (x[j] + x[j - 1]) * half; ++i; ++j; }
                } else {
                    for (int l = 1; l <= mk1; ++l) { t[i] = x[j]; ++i; ++j; }
                }
            }
            place_interp_knots = false;
        }
        poly = (n == nmin);
        nrint = n - nmin + 1;
        nk1 = n - k1;
        {   // boundary knots
            int i = n;
            for (int j = 1; j <= k1; ++j) { t[j] = xb; t[i] = xe; --i; }
        }
        // least-squares spline for the current knots: observation matrix rotated row by row
        fp = 0.0;
        for (int i = 1; i <= nk1; ++i) {
            z[i] = 0.0;
            for (int j = 1; j <= k1; ++j) a[i][j] = 0.0;
        }
        int l = k1;
        for (int it = 1; it <= m; ++it) {
            const double xi = x[it];
            double yi = y[it];
            while (!(xi < t[l + 1] || l == nk1)) l = l + 1;
            sf_bspl(t, k, xi, l, h);
            for (int i = 1; i <= k1; ++i) q[it][i] = h[i];
            int j = l - k1;
            for (int i = 1; i <= k1; ++i) {
                j = j + 1;
                const double piv = h[i];
                if (piv == 0.0) continue;
                double cs, sn;
                sf_givs(piv, a[j][1], cs, sn);
                sf_rota(cs, sn, yi, z[j]);
                if (i == k1) break;
                int i2 = 1;
                for (int i1 = i + 1; i1 <= k1; ++i1) {
                    i2 = i2 + 1;
                    sf_rota(cs, sn, h[i1], a[j][i2]);
                }
            }
            fp = fp + yi * yi;
        }
        if (poly) fp0 = fp;
        fpint[n] = fp0;
        fpint[n - 1] = fpold;
        nrdata[n] = nplus;
        sf_back(a, z, nk1, k1, c);
        fpms = fp - s;
        if ((fpms < 0 ? -fpms : fpms) < acc) goto done;      // acceptable as it is
        if (fpms < 0.0) break;                               // knots accepted -> smoothing
        if (n == nmax) { interp = true; goto done; }         // interpolating spline
        if (n == nest) goto done;                            // storage limit (cannot happen: nest = nmax)
        // how many knots to add
        if (first) {
            nplus = 1;
            first = false;
        } else {
            int npl1 = nplus * 2;
            const double rn = (double)nplus;
            if (fpold - fp > acc) npl1 = (int)(rn * fpms / (fpold - fp));
            int mx = npl1;
            if (nplus / 2 > mx) mx = nplus / 2;
            if (1 > mx) mx = 1;
            nplus = (nplus * 2 < mx) ? nplus * 2 : mx;
        }
        fpold = fp;
        // residual sum of squares per knot interval
        {
            double fpart = 0.0;
            int i = 1, ll = k2, nw = 0;
            for (int it = 1; it <= m; ++it) {
                if (!(x[it] < t[ll] || ll > nk1)) { nw = 1; ll = ll + 1; }
                double term = 0.0;
                int l0 = ll - k2;
                for (int j = 1; j <= k1; ++j) { l0 = l0 + 1; term = term + c[l0] * q[it][j]; }
                term = (term - y[it]) * (term - y[it]);
                fpart = fpart + term;
                if (nw == 0) continue;
                const double store = term * half;
                fpint[i] = fpart - store;
                i = i + 1;
                fpart = store;
                nw = 0;
            }
            fpint[nrint] = fpart;
        }
        for (int lp = 1; lp <= nplus; ++lp) {
            // fpknot: split the interval with the largest residual that still contains data
            {
                const int kk = (n - nrint - 1) / 2;
                double fpmax = 0.0;
                int jbegin = 1, number = 0, maxpt = 0, maxbeg = 0;
                for (int j = 1; j <= nrint; ++j) {
                    const int jpoint = nrdata[j];
                    if (!(fpmax >= fpint[j] || jpoint == 0)) {
                        fpmax = fpint[j]; number = j; maxpt = jpoint; maxbeg = jbegin;
                    }
                    jbegin = jbegin + jpoint + 1;
                }
                const int ihalf = maxpt / 2 + 1;
                const int nrx = maxbeg + ihalf;
                const int next = number + 1;
                if (next <= nrint) {
                    for (int j = next; j <= nrint; ++j) {
                        const int jj = next + nrint - j;
                        fpint[jj + 1] = fpint[jj];
                        nrdata[jj + 1] = nrdata[jj];
                        const int jk = jj + kk;
                        t[jk + 1] = t[jk];
                    }
                }
                nrdata[number] = ihalf - 1;
                nrdata[next] = maxpt - ihalf;
                const double am = (double)maxpt;
                double an = (double)nrdata[number];
                fpint[number] = fpmax * an / am;
                an = (double)nrdata[next];
                fpint[next] = fpmax * an / am;
                const int jk = next + kk;
                t[jk] = x[nrx];
                n = n + 1;
                nrint = nrint + 1;
            }
            if (n == nmax) { place_interp_knots = true; break; }
            if (n == nest) break;
        }
    }
    // ---- part 2: smoothing spline for the accepted knots ------------------------------------
    if (!poly) {
        // discontinuity jumps of the k-th derivative at the interior knots (fpdisc)
        {
            const int nrintd = nk1 - k;
            const double an = (double)nrintd;
            const double fac = an / (t[nk1 + 1] - t[k1]);
            double hd[2 * (SF_MAX_K + 2) + 2];
            for (int l = k2; l <= nk1; ++l) {
                const int lmk = l - k1;
                for (int j = 1; j <= k1; ++j) {
                    const int ik = j + k1, lj = l + j, lk = lj - k2;
                    hd[j] = t[l] - t[lk];
                    hd[ik] = t[l] - t[lj];
                }
                int lp = lmk;
                for (int j = 1; j <= k2; ++j) {
                    int jk = j;
                    double prod = hd[j];
                    for (int i = 1; i <= k; ++i) { jk = jk + 1; prod = prod * hd[jk] * fac; }
                    const int lk = lp + k1;
                    b[lmk][j] = (t[lk] - t[lp]) / prod;
                    lp = lp + 1;
                }
            }
        }
        double p1 = 0.0, f1 = fp0 - s, p3 = -1.0, f3 = fpms, p = 0.0;
        for (int i = 1; i <= nk1; ++i) p = p + a[i][1];
        const double rn = (double)nk1;
        p = rn / p;
        int ich1 = 0, ich3 = 0;
        const int n8 = n - nmin;
        const int n7 = nk1 - k1;
        for (int iter = 1; iter <= maxit; ++iter) {
            const double pinv = 1.0 / p;
            for (int i = 1; i <= nk1; ++i) {
                c[i] = z[i];
                g[i][k2] = 0.0;
                for (int j = 1; j <= k1; ++j) g[i][j] = a[i][j];
            }
            for (int it = 1; it <= n8; ++it) {
                for (int i = 1; i <= k2; ++i) h[i] = b[it][i] * pinv;
                double yi = 0.0;
                for (int j = it; j <= nk1; ++j) {
                    const double piv = h[1];
                    double cs, sn;
                    sf_givs(piv, g[j][1], cs, sn);
                    sf_rota(cs, sn, yi, c[j]);
                    if (j == nk1) break;
                    int i2 = k1;
                    if (j > n7) i2 = nk1 - j;
                    for (int i = 1; i <= i2; ++i) {
                        const int i1 = i + 1;
                        sf_rota(cs, sn, h[i1], g[j][i1]);
                        h[i] = h[i1];
                    }
                    h[i2 + 1] = 0.0;
                }
            }
            sf_back(g, c, nk1, k2, c);
            // f(p)
            fp = 0.0;
            int l = k2;
            for (int it = 1; it <= m; ++it) {
                if (!(x[it] < t[l] || l > nk1)) l = l + 1;
                int l0 = l - k2;
                double term = 0.0;
                for (int j = 1; j <= k1; ++j) { l0 = l0 + 1; term = term + c[l0] * q[it][j]; }
                fp = fp + (term - y[it]) * (term - y[it]);
            }
            fpms = fp - s;
            if ((fpms < 0 ? -fpms : fpms) < acc) break;
            if (iter == maxit) break;
            const double p2 = p, f2 = fpms;
            if (ich3 == 0) {
                if ((f2 - f3) > acc) {
                    if (f2 < 0.0) ich3 = 1;
                } else {
                    p3 = p2; f3 = f2;
                    p = p * con4;
                    if (p <= p1) p = p1 * con9 + p2 * con1;
                    continue;
                }
            }
            if (ich1 == 0) {
                if ((f1 - f2) > acc) {
                    if (f2 > 0.0) ich1 = 1;
                } else {
                    p1 = p2; f1 = f2;
                    p = p / con4;
                    if (p3 < 0.0) continue;
                    if (p >= p3) p = p2 * con1 + p3 * con9;
                    continue;
                }
            }
            if (f2 >= f1 || f2 <= f3) break;      // not monotone: FITPACK returns ier = 2 with this fit
            p = sf_rati(p1, f1, p2, f2, p3, f3);
        }
    }
done:
    (void)interp;
    // ---- evaluation (splev, ext = 3: arguments outside [t(k+1), t(n-k)] are clamped) ----------
    {
        const int nk1e = n - k1;
        const double tb = t[k1], te = t[nk1e + 1];
        int l = k1;
        for (int i = 0; i < n_out; ++i) {
            double arg = (n_out == 1) ? 0.0 : (double)i * ((double)m / (double)(n_out - 1));
            if (i == n_out - 1 && n_out > 1) arg = (double)m;
            if (arg < tb) arg = tb;
            if (arg > te) arg = te;
            while (!(arg < t[l + 1] || l == nk1e)) l = l + 1;
            sf_bspl(t, k, arg, l, h);
            double sp = 0.0;
            int ll = l - k1;
            for (int j = 1; j <= k1; ++j) { ll = ll + 1; sp = sp + c[ll] * h[j]; }
            out[(long long)i * stride] = (OutT)sp;
        }
    }
    return n;
}

}  // namespace m3
