// hostsim.cpp -- TEST INFRASTRUCTURE: the kernel templates of mallorn-astrophysics_amd/csrc
// compiled for the host with the one-lane WaveHost policy (reductions are identities), so the
// feature arithmetic can be checked against the oracle in the GPU-less build container and run
// under -fsanitize=address,undefined.  Never loaded by the product package.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "../../mallorn-astrophysics_amd/csrc/feature_sets.hpp"

using namespace lcfe;

#ifndef HOSTSIM_CAP
#define HOSTSIM_CAP 2048
#endif

extern "C" int hostsim_max_points() { return HOSTSIM_CAP; }

template <int SET>
static void run_all(int64_t n_obj, const int64_t* offsets, const double* t, const double* flux,
                    const double* err, const uint8_t* band, const double* z, double* out,
                    int32_t* status) {
    using W = WaveHost;
    auto ws = std::make_unique<SetLds<SET, HOSTSIM_CAP>>();
    const int ncol = set_ncols(SET);
    const int nst = set_nstatus(SET);
    for (int64_t i = 0; i < n_obj; ++i) {
        const int64_t s = offsets[i];
        const int n = (int)(offsets[i + 1] - s);
        double* row = out + i * ncol;
        int32_t* st = (status && nst) ? status + i * nst : nullptr;
        if (n > HOSTSIM_CAP) {
            fill_row_nan<W>(row, ncol);
            for (int k = 0; k < nst; ++k) if (st) st[k] = -100;
            continue;
        }
        ObjIn in{t + s, flux + s, err + s, band + s, n, z ? z[i] : qnan()};
        RunSet<W, SET, HOSTSIM_CAP>::run(in, *ws, row, st);
    }
}

// the GP kernel has its own working set (Gram matrix) and reads the CSR slice directly; like the
// device it uses the 8-wide sweep for short light curves and the 16-wide one for long ones
static void run_gp(int64_t n_obj, const int64_t* offsets, const double* t, const double* flux, const double* err,
                   const uint8_t* band, double* out, int32_t* status) {
    using W = WaveHost;
    constexpr int NS = 176, NL = 512;      // NL = 512 exercises the 16-wide sweep
    auto ws = std::make_unique<GpLds<NS, 1>>();
    auto wl = std::make_unique<GpLds<NL, 1>>();
    std::vector<double> K((size_t)gp_store_doubles(NL));
    for (int64_t i = 0; i < n_obj; ++i) {
        const int64_t s = offsets[i];
        const int n = (int)(offsets[i + 1] - s);
        ObjIn in{t + s, flux + s, err + s, band + s, n, qnan()};
        int32_t* st = status ? status + set_nstatus(SET_GP2D) * i : nullptr;
        const double* o;
        // like the device: the small working set for short light curves, the large one otherwise
        if (n + 1 <= NS) {
            gp_object<W, NS>(in, *ws, [&](const double* x, int nn, double& f, double* g, bool need) {
                gp_eval<W, NS, double*>(x, nn, *ws, K.data(), f, g, need); }, st);
            o = ws->out;
        } else {
            gp_object<W, NL>(in, *wl, [&](const double* x, int nn, double& f, double* g, bool need) {
                gp_eval<W, NL, double*>(x, nn, *wl, K.data(), f, g, need); }, st);
            o = wl->out;
        }
        for (int k = 0; k < GP_NCOL; ++k) out[i * GP_NCOL + k] = o[k];
    }
}

extern "C" int hostsim_extract(int set, int64_t n_obj, const int64_t* offsets, const double* t,
                               const double* flux, const double* err, const uint8_t* band,
                               const double* z, double* out, int32_t* status) {
    switch (set) {
        case SET_STAT: run_all<SET_STAT>(n_obj, offsets, t, flux, err, band, z, out, status); return 0;
        case SET_BAZIN: run_all<SET_BAZIN>(n_obj, offsets, t, flux, err, band, z, out, status); return 0;
        case SET_POWERLAW: run_all<SET_POWERLAW>(n_obj, offsets, t, flux, err, band, z, out, status); return 0;
        case SET_TDE: run_all<SET_TDE>(n_obj, offsets, t, flux, err, band, z, out, status); return 0;
        case SET_COLOR: run_all<SET_COLOR>(n_obj, offsets, t, flux, err, band, z, out, status); return 0;
        case SET_SHAPE: run_all<SET_SHAPE>(n_obj, offsets, t, flux, err, band, z, out, status); return 0;
        case SET_PHYSICS: run_all<SET_PHYSICS>(n_obj, offsets, t, flux, err, band, z, out, status); return 0;
        case SET_RESEARCH: run_all<SET_RESEARCH>(n_obj, offsets, t, flux, err, band, z, out, status); return 0;
        case SET_GP2D: run_gp(n_obj, offsets, t, flux, err, band, out, status); return 0;
        default: return 1;
    }
}

// ---- per-band 1-D GP (csrc/gp1d.hpp): the four bands g, r, i, z of every object, one after the other
#include "../../mallorn-astrophysics_amd/csrc/gp1d.hpp"

extern "C" int hostsim_gp1d(int64_t n_obj, const int64_t* offsets, const double* t, const double* flux, const double* err,
                            const uint8_t* band, double* out, int32_t* status) {
    using W = WaveHost;
    constexpr int NP = 256;
    auto obj = std::make_unique<ObjLds<HOSTSIM_CAP>>();
    auto ws = std::make_unique<Gp1dLds<NP, 1>>();
    std::vector<double> K((size_t)gp_store_doubles(NP));
    for (int64_t i = 0; i < n_obj; ++i) {
        const int64_t s = offsets[i];
        const int n = (int)(offsets[i + 1] - s);
        double* o = out + i * GP1D_NCOL;
        for (int k = 0; k < GP1D_NCOL; ++k) o[k] = qnan();
        if (n > HOSTSIM_CAP) continue;
        ObjIn in{t + s, flux + s, err + s, band + s, n, qnan()};
        stage_object<W, HOSTSIM_CAP>(in, *obj);
        bool fitted[4];
        for (int j = 0; j < 4; ++j) {
            const int b = j + 1;                                   // g, r, i, z
            const int bs = obj->boff[b], m = obj->boff[b + 1] - bs;
            fitted[j] = m >= 5;
            gp1d_band<W, NP>([&](int r, double& tt, double& ff, double& ee) { tt = obj->bt[bs + r]; ff = obj->bf[bs + r]; ee = obj->be[bs + r]; }, m, *ws,
                             [&](const double* x, int nn, double& f, double* g) { gp1d_eval<W, NP, double*>(x, nn, *ws, K.data(), f, g); },
                             o + 4 * j, status ? status + i * GP1D_NSTATUS + j : nullptr);
        }
        gp1d_cross_band(o, fitted);
    }
    return 0;
}

// ---- bounded L-BFGS-B (csrc/lbfgsb_box.hpp) with the objective supplied by the caller: the test
// drives it with the same Python function it hands to scipy.optimize.minimize.
#include "../../mallorn-astrophysics_amd/csrc/lbfgsb_box.hpp"

typedef void (*hostsim_fg_t)(const double* x, double* f, double* g);

extern "C" int hostsim_lbfgsb_box3(double* x, const double* lo, const double* hi, hostsim_fg_t fg, int maxiter,
                                   int maxfun, double factr, double pgtol, double* f_out, int* n_iter, int* n_eval,
                                   double* trace, int trace_cap) {
    double Sm[10][3], Ym[10][3];
    double xx[3] = {x[0], x[1], x[2]}, f = 0;
    int nt = 0;
    auto ev = [&](const double* p, double& fv, double* gv) {
        fg(p, &fv, gv);
        if (trace && nt < trace_cap) { trace[4 * nt] = p[0]; trace[4 * nt + 1] = p[1]; trace[4 * nt + 2] = p[2]; trace[4 * nt + 3] = fv; ++nt; }
    };
    const int rc = lcfe::lbfgsb_box_minimize<3, 10>(xx, lo, hi, f, ev, maxiter, maxfun, factr, pgtol, 20, *n_iter, *n_eval, Sm, Ym);
    for (int i = 0; i < 3; ++i) x[i] = xx[i];
    *f_out = f;
    return rc;
}
