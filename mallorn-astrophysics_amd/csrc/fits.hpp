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

// bazin_fitting.py:63-179 for one band's time-sorted rows -> out8 (wave-shared, lane 0 writes)
template <class W, class Store>
LCFE_FN TrfResult bazin_fit_band(const double* t, const double* f, const double* e, int m,
                                 Store& V, double* slot, unsigned long long* keys, double* out8) {
    const int lane = W::lane();
    TrfResult res{TRF_FAIL_TOO_FEW, 0};
    if (m < 5) {                                                   // :76-87
        if (lane == 0) for (int k = 0; k < 8; ++k) out8[k] = qnan();
        return res;
    }
    const int pk = wave_argmax_first<W>(f, m);                     // :97  np.argmax on the sorted rows
    const double med = wave_median<W>(f, m, slot, keys);           // :99-100
    const double fpk = f[pk];
    const double duration = t[m - 1] - t[0];                       // :103
    double mx = -__builtin_inf();
    bool nanf = false;
    for (int i = lane; i < m; i += W::LANES) { mx = fmax(mx, f[i]); nanf = nanf || is_nan(f[i]); }
    mx = W::max(mx);
    if (W::any(nanf)) mx = qnan();
    Vec<5> x, lb, ub;
    x[0] = fpk - med; x[1] = t[pk]; x[2] = duration * 0.2; x[3] = duration * 0.3; x[4] = med;    // :132
    lb[0] = 0; lb[1] = t[0]; lb[2] = 0.1; lb[3] = 0.1; lb[4] = -mx;                                // :114-118
    ub[0] = 3 * mx; ub[1] = t[m - 1]; ub[2] = duration; ub[3] = duration; ub[4] = 2 * mx;
    for (int i = lane; i < m; i += W::LANES) {
        const double sg = (e[i] > 0) ? e[i] : 1.0;                 // :126
        V.w[i] = 1.0 / sg;                                         // _minpack_py.py:981 transform = 1/sigma
    }
    W::sync();
    res = trf_fit<W, BazinModel, Store>(BazinModel(), t, f, m, x, lb, ub, 2000, V);
    if (res.status <= 0) {                                         // :168-179 any exception -> NaN
        if (lane == 0) for (int k = 0; k < 8; ++k) out8[k] = qnan();
        return res;
    }
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
    return res;
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
    if (WW::lane() == 0) {
        double* o = S.out;
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
    WW::sync();
}

// ---------------------------------------------------------------- post-peak decline models
// train_v55_powerlaw.py:108-133.  pow(max(t - t0, 0.1), p): the kink makes the FD Jacobian part of
// the semantics, so the models are evaluated exactly as written.
template <int ID>
struct DeclineModel;

template <int ID>
LCFE_FN constexpr double decline_exponent() {
    return ID == 0 ? -5.0 / 3.0 : ID == 1 ? -1.0 : ID == 2 ? -1.5 : ID == 3 ? -2.0 : ID == 4 ? -2.5 : ID == 5 ? -3.0 : -0.5;
}

template <int ID>
struct DeclineModel {          // ID 0..6: A * max(t - t0, 0.1)^p     p = (A, t0)
    static constexpr int NP = 2;
    LCFE_FN double operator()(double t, const Vec<2>& p) const {
        return p[0] * pow(fmax(t - p[1], 0.1), decline_exponent<ID>());
    }
};
template <>
struct DeclineModel<7> {       // exponential: A * exp(-max(t - t0, 0) / tau)   p = (A, tau, t0)
    static constexpr int NP = 3;
    LCFE_FN double operator()(double t, const Vec<3>& p) const { return p[0] * exp(-fmax(t - p[2], 0.0) / p[1]); }
};
template <>
struct DeclineModel<8> {       // linear: A - b * max(t - t0, 0)               p = (A, b, t0)
    static constexpr int NP = 3;
    LCFE_FN double operator()(double t, const Vec<3>& p) const { return p[0] - p[1] * fmax(t - p[2], 0.0); }
};

template <int CAP>
struct PowerlawLds {
    union {                      // the n = 2 and n = 3 fits run one after the other in the same block
        TrfLds<3, CAP> t3;
        TrfLds<2, CAP> t2;
    } trf;
    double tp[CAP];              // post-peak times relative to the peak
    double fp[CAP];              // post-peak fluxes
    double out[POWERLAW_NCOL];
};

template <int N, int CAP>
LCFE_FN TrfLds<N, CAP>& trf_block(PowerlawLds<CAP>& S) {
    if constexpr (N == 2) return S.trf.t2;
    else return S.trf.t3;
}

template <class W, int ID, int CAP>
LCFE_FN TrfResult decline_fit(int k, double peak_flux, double mean_post, double ss_tot, PowerlawLds<CAP>& S,
                              double* out) {
    using M = DeclineModel<ID>;
    constexpr int N = M::NP;
    const int lane = W::lane();
    Vec<N> x, lb, ub;
    if (N == 2) {                                                 // :172-175
        x[0] = peak_flux; x[1] = 0;
        lb[0] = 0; lb[1] = -10; ub[0] = 1e6; ub[1] = 10;
    } else if (ID == 7) {                                         // :177-180
        x[0] = peak_flux; x[1] = 30; x[N - 1] = 0;
        lb[0] = 0; lb[1] = 1; lb[N - 1] = -10; ub[0] = 1e6; ub[1] = 500; ub[N - 1] = 10;
    } else {                                                      // :181-184
        x[0] = peak_flux; x[1] = 1; x[N - 1] = 0;
        lb[0] = 0; lb[1] = 0; lb[N - 1] = -10; ub[0] = 1e6; ub[1] = 100; ub[N - 1] = 10;
    }
    TrfLds<N, CAP>& T = trf_block<N>(S);
    for (int i = lane; i < k; i += W::LANES) T.w[i] = 1.0;        // unweighted: r = model - y
    W::sync();
    M model;
    TrfResult res = trf_fit<W, M, TrfLds<N, CAP>>(model, S.tp, S.fp, k, x, lb, ub, 1000, T);
    if (res.status <= 0) {                                        // :191-192
        if (lane == 0) *out = qnan();
        return res;
    }
    double ss = 0;
    for (int i = lane; i < k; i += W::LANES) { const double r = S.fp[i] - model(S.tp[i], x); ss += r * r; }   // :186-187
    ss = W::sum(ss);
    if (lane == 0) *out = (ss_tot > 0) ? 1.0 - ss / ss_tot : 0.0;       // :189
    return res;
}

// train_v55_powerlaw.py:147-194 for one band (time-sorted rows) -> 9 R^2 values
template <class W, int CAP>
LCFE_FN void decline_band(const double* t, const double* f, int m, PowerlawLds<CAP>& S, double* out9,
                          int32_t* st) {
    const int lane = W::lane();
    auto fail_all = [&](int code) {
        if (lane == 0) for (int j = 0; j < 9; ++j) out9[j] = qnan();
        if (st && lane == 0) for (int j = 0; j < 9; ++j) { st[2 * j] = code; st[2 * j + 1] = 0; }
    };
    if (m < 5) { fail_all(TRF_FAIL_TOO_FEW); return; }            // :150-151
    const int pk = wave_argmax_first<W>(f, m);                    // :157
    const double peak_time = t[pk], peak_flux = f[pk];
    // post-peak rows: t > peak_time (:161); rows are time-sorted, so they form a suffix -- but
    // equal time stamps after the peak are excluded by the strict comparison, so count explicitly
    int first = m;
    for (int i = lane; i < m; i += W::LANES) if (t[i] > peak_time) first = (i < first) ? i : first;
    first = W::min(first);
    const int k = m - first;
    if (k < 3) { fail_all(TRF_FAIL_TOO_FEW); return; }            // :162-163
    double sum = 0;
    for (int i = lane; i < k; i += W::LANES) {
        S.tp[i] = t[first + i] - peak_time;                       // :165
        S.fp[i] = f[first + i];
        sum += f[first + i];
    }
    sum = W::sum(sum);
    const double mean_post = sum / k;
    W::sync();
    double ss_tot = 0;
    for (int i = lane; i < k; i += W::LANES) { const double d = S.fp[i] - mean_post; ss_tot += d * d; }    // :188
    ss_tot = W::sum(ss_tot);
    TrfResult r;
#define LCFE_DECLINE(ID)                                                                  \
    r = decline_fit<W, ID, CAP>(k, peak_flux, mean_post, ss_tot, S, out9 + ID);           \
    if (st && lane == 0) { st[2 * ID] = r.status; st[2 * ID + 1] = r.nfev; }             \
    W::sync();
    LCFE_DECLINE(0) LCFE_DECLINE(1) LCFE_DECLINE(2) LCFE_DECLINE(3) LCFE_DECLINE(4)
    LCFE_DECLINE(5) LCFE_DECLINE(6) LCFE_DECLINE(7) LCFE_DECLINE(8)
#undef LCFE_DECLINE
}

template <class W, int CAP>
LCFE_FN void powerlaw_object(const ObjLds<CAP>& L, PowerlawLds<CAP>& S, int32_t* st) {
    for (int j = 0; j < 3; ++j) {                                 // :198 bands g, r, i
        const int kb = j + 1;
        const int s = L.boff[kb], m = L.boff[kb + 1] - s;
        decline_band<W, CAP>(L.bt + s, L.bf + s, m, S, S.out + 9 * j, st ? st + 18 * j : nullptr);
        W::sync();
    }
}

}  // namespace lcfe
