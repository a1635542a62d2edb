// fits.hpp -- the bounded curve fits of the feature frame: Bazin (bazin_fitting.py) and the nine
// post-peak decline models of the v55 set (scripts/train_v55_powerlaw.py:108-202).
#pragma once
#include "stage.hpp"
#include "trf.hpp"

namespace lcfe {

constexpr int BAZIN_NCOL = 52;
constexpr int POWERLAW_NCOL = 27;

// np.clip with NaN propagation
LCFE_FN double np_clip(double x, double lo, double hi) { return is_nan(x) ? x : fmin(fmax(x, lo), hi); }

// np.median of m wave-shared values (m >= 1) by rank counting; uniform result.  `slot` = 2 doubles,
// `keys` = m words of wave-shared scratch.
template <class W>
LCFE_FN double wave_median(const double* x, int m, double* slot, unsigned long long* keys) {
    const int ranks[2] = {(m - 1) / 2, m / 2};
    bool nanf = false;
    for (int i = W::lane(); i < m; i += W::LANES) nanf = nanf || is_nan(x[i]);
    wave_select_ranks<W, 2>(x, m, keys, ranks, slot);
    const double med = (ranks[0] == ranks[1]) ? slot[0] : (slot[0] + slot[1]) / 2.0;
    const bool any_nan = W::any(nanf);
    W::sync();
    return any_nan ? qnan() : med;
}

// Values of ranks lo and hi among n wave-shared values -> slot[0], slot[1]
template <class W>
LCFE_FN void wave_rank_select(const double* x, int n, int lo, int hi, double* slot, unsigned long long* keys) {
    const int ranks[2] = {lo, hi};
    wave_select_ranks<W, 2>(x, n, keys, ranks, slot);
}

// ---------------------------------------------------------------- Bazin
struct BazinModel {
    static constexpr int NP = 5;
    // bazin_fitting.py:37-60   p = (A, t0, tau_rise, tau_fall, B)
    LCFE_FN double operator()(double t, const Vec<5>& p) const {
        const double numerator = exp(-(t - p[1]) / p[3]);
        const double denominator = 1.0 + exp(-(t - p[1]) / p[2]);
        return p[0] * numerator / denominator + p[4];
    }
    // The finite-difference Jacobian evaluates the model at x + h e_k for k = 0..4 (trf.hpp::trf_jacobian).  The two
    // exponentials of a row depend on (t0, tau_fall) resp. (t0, tau_rise) only, so a step in A or B leaves both and a
    // step in one time scale leaves one of them bit for bit what they are at x -- and those were computed by the
    // residual evaluation at x that precedes every Jacobian (kept in the matrix columns the QR has left free): four
    // exponentials per row instead of ten, the same values as operator() gives.
    static constexpr int kSharedTerms = 2;
    LCFE_FN void terms(double t, const Vec<5>& p, double (&q)[2]) const {
        q[0] = exp(-(t - p[1]) / p[3]);
        q[1] = 1.0 + exp(-(t - p[1]) / p[2]);
    }
    LCFE_FN double value(const Vec<5>& p, const double (&q)[2]) const { return p[0] * q[0] / q[1] + p[4]; }
    template <int K>
    LCFE_FN double stepped(double t, const Vec<5>& p1, const double (&at_x)[2]) const {
        const double numerator = (K == 1 || K == 3) ? exp(-(t - p1[1]) / p1[3]) : at_x[0];
        const double denominator = (K == 1 || K == 2) ? 1.0 + exp(-(t - p1[1]) / p1[2]) : at_x[1];
        return p1[0] * numerator / denominator + p1[4];
    }
};

// The six band fits of one light curve share one pool: band k (rows boff[k]..boff[k+1]) owns the
// rows boff[k] + 5k .. of every column, so the fits can run side by side in different lane groups.
template <int CAP>
struct BazinLds {
    double A[6][CAP + 30];
    double r[CAP], rn[CAP], w[CAP];
    unsigned long long keys[CAP];
    double slot[8][2];
    double out[BAZIN_NCOL];
};

template <int CAP>
LCFE_FN TrfView<5> bazin_view(BazinLds<CAP>& S, int band, int band_start) {
    TrfView<5> v;
#pragma unroll
    for (int k = 0; k < 6; ++k) v.A[k] = S.A[k] + band_start + 5 * band;
    v.r = S.r + band_start;
    v.rn = S.rn + band_start;
    v.w = S.w + band_start;
    return v;
}

// bazin_fitting.py:97-126: start point, bounds and weights of one band's fit (m >= 5 time-sorted rows)
template <class W, class Store>
LCFE_FN void bazin_prepare(const double* t, const double* f, const double* e, int m, Store& V, double* slot,
                           unsigned long long* keys, TrfState<5>& Z) {
    const int lane = W::lane();
    const int pk = wave_argmax_first<W>(f, m);                     // :97  np.argmax on the sorted rows
    const double med = wave_median<W>(f, m, slot, keys);           // :99-100
    const double fpk = f[pk];
    const double duration = t[m - 1] - t[0];                       // :103
    double mx = -__builtin_inf();
    bool nanf = false;
    for (int i = lane; i < m; i += W::LANES) { mx = fmax(mx, f[i]); nanf = nanf || is_nan(f[i]); }
    mx = W::max(mx);
    if (W::any(nanf)) mx = qnan();
    Vec<5>& x = Z.x; Vec<5>& lb = Z.lb; Vec<5>& ub = Z.ub;
    x[0] = fpk - med; x[1] = t[pk]; x[2] = duration * 0.2; x[3] = duration * 0.3; x[4] = med;    // :132
    lb[0] = 0; lb[1] = t[0]; lb[2] = 0.1; lb[3] = 0.1; lb[4] = -mx;                                // :114-118
    ub[0] = 3 * mx; ub[1] = t[m - 1]; ub[2] = duration; ub[3] = duration; ub[4] = 2 * mx;
    for (int i = lane; i < m; i += W::LANES) {
        const double sg = (e[i] > 0) ? e[i] : 1.0;                 // :126
        V.w[i] = 1.0 / sg;                                         // _minpack_py.py:981 transform = 1/sigma
    }
    Z.max_nfev = 2000;                                             // :136 maxfev
    W::sync();
}

// bazin_fitting.py:139-179: clipping, reduced chi^2 and the derived columns of a finished fit -> out8
template <class W>
LCFE_FN void bazin_finish(const double* t, const double* f, const double* e, int m, const TrfState<5>& Z, double* out8) {
    const int lane = W::lane();
    if (Z.res.status <= 0) {                                       // :168-179 any exception -> NaN
        if (lane == 0) for (int k = 0; k < 8; ++k) out8[k] = qnan();
        return;
    }
    const Vec<5>& x = Z.x;
    Vec<5> q;
    q[0] = np_clip(x[0], -1e6, 1e6);                               // :142-145
    q[1] = x[1];
    q[2] = np_clip(x[2], 0.1, 1e4);
    q[3] = np_clip(x[3], 0.1, 1e4);
    q[4] = np_clip(x[4], -1e6, 1e6);
    double chi2 = 0;
    BazinModel model;
    for (int i = lane; i < m; i += W::LANES) {                     // :148-150
        const double sg = (e[i] > 0) ? e[i] : 1.0;
        const double r = (f[i] - model(t[i], q)) / sg;
        chi2 += r * r;
    }
    chi2 = W::sum(chi2);
    if (lane == 0) {
        out8[0] = q[0]; out8[1] = q[1]; out8[2] = q[2]; out8[3] = q[3]; out8[4] = q[4];
        out8[5] = np_clip(chi2 / (double)(m - 5), 0.0, 1e6);       // :151 (m == 5: x/0 -> inf -> 1e6)
        out8[6] = np_clip(q[2] / (q[3] + 1e-6), 0.0, 100.0);       // :154
        out8[7] = np_clip(q[0] + q[4], -1e6, 1e6);                 // :155
    }
}

// bazin_fitting.py:63-179 for one band's time-sorted rows -> out8 (wave-shared, lane 0 writes)
template <class W, class Store>
LCFE_FN TrfResult bazin_fit_band(const double* t, const double* f, const double* e, int m,
                                 Store& V, double* slot, unsigned long long* keys, double* out8) {
    const int lane = W::lane();
    if (m < 5) {                                                   // :76-87
        if (lane == 0) for (int k = 0; k < 8; ++k) out8[k] = qnan();
        return TrfResult{TRF_FAIL_TOO_FEW, 0};
    }
    TrfState<5> Z;
    bazin_prepare<W, Store>(t, f, e, m, V, slot, keys, Z);
    trf_begin<W, BazinModel, Store>(BazinModel(), t, f, m, Z, V);
    while (Z.phase != TRF_PH_DONE) {
        if (Z.phase == TRF_PH_OUTER) trf_outer<W, BazinModel, Store>(m, Z, V);
        else trf_inner<W, BazinModel, Store>(BazinModel(), t, f, m, Z, V);
    }
    bazin_finish<W>(t, f, e, m, Z, out8);
    return Z.res;
}

// bazin_fitting.py:217-249: the four cross-band columns from the 48 per-band ones (scalar code)
LCFE_FN void mean_std_small(const double* v, int n, double& mean, double& sd);
LCFE_FN void bazin_cross_band(double* o) {
    double v[6];
    int n = 0;
    double mean, sd;
    for (int k = 1; k <= 3; ++k) if (!is_nan(o[8 * k + 2])) v[n++] = o[8 * k + 2];     // :217-225 g,r,i
    if (n >= 2) { mean_std_small(v, n, mean, sd); o[48] = sd / mean; } else o[48] = qnan();
    n = 0;
    for (int k = 1; k <= 3; ++k) if (!is_nan(o[8 * k + 3])) v[n++] = o[8 * k + 3];
    if (n >= 2) { mean_std_small(v, n, mean, sd); o[49] = sd / mean; } else o[49] = qnan();
    n = 0;
    for (int k = 0; k < 6; ++k) if (!is_nan(o[8 * k + 5])) v[n++] = o[8 * k + 5];       // :238-249
    if (n > 0) { mean_std_small(v, n, mean, sd); o[50] = mean; o[51] = sd; } else { o[50] = qnan(); o[51] = qnan(); }
}

// np.std(v)/np.mean(v) style helpers over <= 6 values (lane-0 scalar code)
LCFE_FN void mean_std_small(const double* v, int n, double& mean, double& sd) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += v[i];
    mean = s / n;
    double q = 0;
    for (int i = 0; i < n; ++i) { const double d = v[i] - mean; q += d * d; }
    sd = sqrt(q / n);
}

// bazin_fitting.py:182-251
// W: the policy of ONE FIT (a whole wave, or an 8-lane group: then the six bands run side by side
// in six groups of the wave).  WW: the policy of the whole wave (cross-band epilogue).
template <class W, class WW, int CAP>
LCFE_FN void bazin_object(const ObjLds<CAP>& L, BazinLds<CAP>& S, int32_t* st) {
    for (int k = W::group_id(); k < 6; k += W::NGROUPS) {
        const int s = L.boff[k], m = L.boff[k + 1] - s;
        TrfView<5> V = bazin_view(S, k, s);
        TrfResult r = bazin_fit_band<W, TrfView<5>>(L.bt + s, L.bf + s, L.be + s, m, V, S.slot[W::group_id()], S.keys + s, S.out + 8 * k);
        if (st && W::lane() == 0) { st[2 * k] = r.status; st[2 * k + 1] = r.nfev; }
        W::sync();
    }
    WW::sync();
    if (WW::lane() == 0) bazin_cross_band(S.out);
    WW::sync();
}

// ---------------------------------------------------------------- post-peak decline models
// train_v55_powerlaw.py:108-133.  pow(max(t - t0, 0.1), p): the kink makes the FD Jacobian part of
// the semantics, so the models are evaluated exactly as written.  The seven power laws differ only
// in the exponent, which is DATA here: fits of different power laws then share one instruction
// stream and can run side by side in different lane groups.
struct PowerModel {            // A * max(t - t0, 0.1)^p     x = (A, t0)
    static constexpr int NP = 2;
    double p;
    LCFE_FN double operator()(double t, const Vec<2>& x) const { return x[0] * pow(fmax(t - x[1], 0.1), p); }
    // a step in A leaves the power bit for bit what the residual evaluation at x computed (see BazinModel)
    static constexpr int kSharedTerms = 1;
    LCFE_FN void terms(double t, const Vec<2>& x, double (&q)[1]) const { q[0] = pow(fmax(t - x[1], 0.1), p); }
    LCFE_FN double value(const Vec<2>& x, const double (&q)[1]) const { return x[0] * q[0]; }
    template <int K>
    LCFE_FN double stepped(double t, const Vec<2>& x1, const double (&at_x)[1]) const {
        return x1[0] * ((K == 1) ? pow(fmax(t - x1[1], 0.1), p) : at_x[0]);
    }
};
struct ExpModel {              // A * exp(-max(t - t0, 0) / tau)   x = (A, tau, t0)
    static constexpr int NP = 3;
    LCFE_FN double operator()(double t, const Vec<3>& x) const { return x[0] * exp(-fmax(t - x[2], 0.0) / x[1]); }
};
struct LinModel {              // A - b * max(t - t0, 0)           x = (A, b, t0)
    static constexpr int NP = 3;
    LCFE_FN double operator()(double t, const Vec<3>& x) const { return x[0] - x[1] * fmax(t - x[2], 0.0); }
};

LCFE_FN double decline_exponent(int id) {
    // order of MODELS (train_v55_powerlaw.py:135-145): -5/3, -1, -1.5, -2, -2.5, -3, -0.5
    const double e[7] = {-5.0 / 3.0, -1.0, -1.5, -2.0, -2.5, -3.0, -0.5};
    return e[id];
}

// fit units in flight per band: two while the pool fits next to the staged object in 160 KiB of LDS
template <int CAP> struct powerlaw_slots { static constexpr int SL = (CAP <= 512) ? 2 : 1; };

template <int CAP>
struct PowerlawLds {
    static constexpr int SL = powerlaw_slots<CAP>::SL;
    // pool of fit workspaces: every (band, slot) pair owns the rows [off, off + k_band + 3) of each column
    double A[4][SL * CAP + 24];
    double r[SL * CAP + 24], rn[SL * CAP + 24], w[SL * CAP + 24];
    double tp[CAP];              // post-peak times relative to the peak (per band segment)
    double fp[CAP];              // post-peak fluxes
    double peak[3], sstot[3];    // per band: peak flux, total sum of squares of the post-peak fluxes
    int k[3], first[3];          // per band: number of post-peak rows (or -1: no fits), first post-peak row
    double out[POWERLAW_NCOL];
};

template <int N, int CAP>
LCFE_FN TrfView<N> powerlaw_view(PowerlawLds<CAP>& S, int off) {
    TrfView<N> v;
#pragma unroll
    for (int c = 0; c <= N; ++c) v.A[c] = S.A[c] + off;
    v.r = S.r + off;
    v.rn = S.rn + off;
    v.w = S.w + off;
    return v;
}

// exponential and linear decline models behind one type (the three-parameter fit kernel runs both side by side)
struct Decline3Model {
    static constexpr int NP = 3;
    int kind;                  // 7: exponential, 8: linear
    LCFE_FN double operator()(double t, const Vec<3>& x) const {
        if (kind == 7) return x[0] * exp(-fmax(t - x[2], 0.0) / x[1]);
        return x[0] - x[1] * fmax(t - x[2], 0.0);
    }
};

// start point and bounds of a decline fit (train_v55_powerlaw.py:172-184); kind 0..6 power laws, 7 exponential, 8 linear
template <int N>
LCFE_FN void decline_setup(int kind, double peak_flux, TrfState<N>& Z) {
    Vec<N>& x = Z.x; Vec<N>& lb = Z.lb; Vec<N>& ub = Z.ub;
    if (N == 2) {                                                 // :172-175
        x[0] = peak_flux; x[1] = 0;
        lb[0] = 0; lb[1] = -10; ub[0] = 1e6; ub[1] = 10;
    } else if (kind == 7) {                                       // :177-180 exponential
        x[0] = peak_flux; x[1] = 30; x[N - 1] = 0;
        lb[0] = 0; lb[1] = 1; lb[N - 1] = -10; ub[0] = 1e6; ub[1] = 500; ub[N - 1] = 10;
    } else {                                                      // :181-184 linear
        x[0] = peak_flux; x[1] = 1; x[N - 1] = 0;
        lb[0] = 0; lb[1] = 0; lb[N - 1] = -10; ub[0] = 1e6; ub[1] = 100; ub[N - 1] = 10;
    }
    Z.max_nfev = 1000;                                            // maxfev=1000
}

// R^2 of a finished decline fit (:186-192) -> *out
template <class W, class M>
LCFE_FN void decline_finish(const M& model, const double* tp, const double* fp, int k, double ss_tot,
                            const TrfState<M::NP>& Z, double* out) {
    const int lane = W::lane();
    if (Z.res.status <= 0) {                                      // :191-192
        if (lane == 0) *out = qnan();
        return;
    }
    double ss = 0;
    for (int i = lane; i < k; i += W::LANES) { const double r = fp[i] - model(tp[i], Z.x); ss += r * r; }   // :186-187
    ss = W::sum(ss);
    if (lane == 0) *out = (ss_tot > 0) ? 1.0 - ss / ss_tot : 0.0;       // :189
}

// one bounded fit of `model` to the k post-peak rows (tp, fp) -> R^2   (train_v55_powerlaw.py:170-192)
template <class W, class M>
LCFE_FN TrfResult decline_fit(const M& model, const double* tp, const double* fp, int k, double peak_flux,
                              double ss_tot, int kind, TrfView<M::NP>& T, double* out) {
    constexpr int N = M::NP;
    const int lane = W::lane();
    TrfState<N> Z;
    decline_setup<N>(kind, peak_flux, Z);
    for (int i = lane; i < k; i += W::LANES) T.w[i] = 1.0;        // unweighted: r = model - y
    W::sync();
    trf_begin<W, M, TrfView<N>>(model, tp, fp, k, Z, T);
    while (Z.phase != TRF_PH_DONE) {
        if (Z.phase == TRF_PH_OUTER) trf_outer<W, M, TrfView<N>>(k, Z, T);
        else trf_inner<W, M, TrfView<N>>(model, tp, fp, k, Z, T);
    }
    decline_finish<W, M>(model, tp, fp, k, ss_tot, Z, out);
    return Z.res;
}

// train_v55_powerlaw.py:147-166 for one band's time-sorted rows: peak, post-peak rows (times relative to the peak and
// fluxes -> tp, fp), their total sum of squares.  k = -1: no fits for this band.
template <class W>
LCFE_FN void powerlaw_band_prepare(const double* t, const double* f, int m, double* tp, double* fp, int& k, int& first,
                                   double& peak_flux, double& ss_tot) {
    k = -1; first = 0; peak_flux = 0; ss_tot = 0;
    if (m >= 5) {                                             // :150-151
        const int pk = wave_argmax_first<W>(f, m);            // :157
        const double peak_time = t[pk];
        peak_flux = f[pk];
        first = m;
        for (int i = W::lane(); i < m; i += W::LANES) if (t[i] > peak_time) first = (i < first) ? i : first;   // :161
        first = W::min(first);
        k = m - first;
        if (k < 3) k = -1;                                    // :162-163
        else {
            double sum = 0;
            for (int i = W::lane(); i < k; i += W::LANES) {
                tp[i] = t[first + i] - peak_time;             // :165
                fp[i] = f[first + i];
                sum += f[first + i];
            }
            const double mean_post = W::sum(sum) / k;
            W::sync();
            double q = 0;
            for (int i = W::lane(); i < k; i += W::LANES) { const double d = fp[i] - mean_post; q += d * d; }   // :188
            ss_tot = W::sum(q);
        }
    }
}

// W: policy of ONE FIT (an 8-lane group on the device: up to six fits side by side -- three bands x
// two slots); WW: policy of the whole wave.
template <class W, class WW, int CAP>
LCFE_FN void powerlaw_object(const ObjLds<CAP>& L, PowerlawLds<CAP>& S, int32_t* st) {
    const int g = W::group_id();
    // ---- per band (train_v55_powerlaw.py:147-166): peak, post-peak rows, their sum of squares
    for (int j = g; j < 3; j += W::NGROUPS) {
        const int kb = j + 1;
        const int s = L.boff[kb], m = L.boff[kb + 1] - s;
        const double* t = L.bt + s;
        const double* f = L.bf + s;
        int k, first;
        double peak_flux, ss_tot;
        powerlaw_band_prepare<W>(t, f, m, S.tp + s, S.fp + s, k, first, peak_flux, ss_tot);
        if (W::lane() == 0) { S.k[j] = k; S.first[j] = first; S.peak[j] = peak_flux; S.sstot[j] = ss_tot; }
        W::sync();
    }
    WW::sync();
    // ---- 27 fits: unit u = band j (0..2) x slot; with two slots, slot 0 takes the power laws 0,2,4,6
    // and the exponential, slot 1 the power laws 1,3,5 and the linear model
    constexpr int SL = powerlaw_slots<CAP>::SL;
    constexpr int NUNIT = 3 * SL;
    constexpr int NU = (W::NGROUPS >= NUNIT) ? NUNIT : W::NGROUPS;     // units in flight
    for (int u = g; u < NUNIT; u += NU) {
        if (g >= NU) break;
        const int j = u % 3, slot = u / 3;
        const int kb = j + 1;
        const int s = L.boff[kb];
        const int k = S.k[j];
        double* out9 = S.out + 9 * j;
        int32_t* stj = st ? st + 18 * j : nullptr;
        if (k < 0) {
            if (W::lane() == 0) {
                for (int id = slot; id < 9; id += SL) { out9[id] = qnan(); if (stj) { stj[2 * id] = TRF_FAIL_TOO_FEW; stj[2 * id + 1] = 0; } }
            }
            continue;
        }
        // workspace rows of this unit: both slots of a band fit inside 2*(band length) + 6 rows
        const int off = SL * (s - L.boff[1]) + 3 * SL * j + slot * (k + 3);
        const double* tp = S.tp + s;
        const double* fp = S.fp + s;
        for (int id = slot; id < 7; id += SL) {
            TrfView<2> T = powerlaw_view<2, CAP>(S, off);
            PowerModel model{decline_exponent(id)};
            TrfResult r = decline_fit<W, PowerModel>(model, tp, fp, k, S.peak[j], S.sstot[j], id, T, out9 + id);
            if (stj && W::lane() == 0) { stj[2 * id] = r.status; stj[2 * id + 1] = r.nfev; }
            W::sync();
        }
        TrfView<3> T3 = powerlaw_view<3, CAP>(S, off);
        if (slot == 0) {
            TrfResult r = decline_fit<W, ExpModel>(ExpModel(), tp, fp, k, S.peak[j], S.sstot[j], 7, T3, out9 + 7);
            if (stj && W::lane() == 0) { stj[14] = r.status; stj[15] = r.nfev; }
            W::sync();
        }
        if (slot == SL - 1) {
            TrfResult r = decline_fit<W, LinModel>(LinModel(), tp, fp, k, S.peak[j], S.sstot[j], 8, T3, out9 + 8);
            if (stj && W::lane() == 0) { stj[16] = r.status; stj[17] = r.nfev; }
            W::sync();
        }
    }
    WW::sync();
}

}  // namespace lcfe
