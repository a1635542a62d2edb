// feature_sets.hpp -- dispatch of one object to a feature set, shared by the HIP kernels
// (lcfe.hip) and the host simulation (tests/hostsim).
#pragma once
#include "stage.hpp"
#include "stat.hpp"
#include "fits.hpp"
#include "tde.hpp"
#include "color.hpp"
#include "shape.hpp"
#include "physics.hpp"
#include "gp.hpp"
#include "research.hpp"

namespace lcfe {

enum { SET_STAT = 0, SET_BAZIN, SET_POWERLAW, SET_TDE, SET_COLOR, SET_SHAPE, SET_PHYSICS, SET_GP2D, SET_GP1D, SET_RESEARCH, NUM_SETS };

LCFE_HD int set_ncols(int set) {
    switch (set) {
        case SET_STAT: return 123;
        case SET_BAZIN: return 52;
        case SET_POWERLAW: return 27;
        case SET_TDE: return 25;
        case SET_COLOR: return 83;
        case SET_SHAPE: return 65;
        case SET_PHYSICS: return 32;
        case SET_GP2D: return 27;
        case SET_GP1D: return 21;
        case SET_RESEARCH: return 40;
    }
    return 0;
}
LCFE_HD int set_nstatus(int set) {
    switch (set) {
        case SET_BAZIN: return 12;
        case SET_POWERLAW: return 54;
#ifdef LCFE_GP_PROF
        case SET_GP2D: return 16;
#else
        case SET_GP2D: return 4;
#endif
        case SET_GP1D: return 4;
        case SET_RESEARCH: return 1;
    }
    return 0;
}

// Wave-shared working memory (LDS on the device) of one object, per feature set.
template <int SET, int CAP>
struct SetLds;

template <int CAP>
struct SetLds<SET_STAT, CAP> {
    ObjLds<CAP> obj;
    StatScratch<CAP> stat;
};

template <int CAP>
struct SetLds<SET_BAZIN, CAP> {
    ObjLds<CAP> obj;
    BazinLds<CAP> fit;
};

template <int CAP>
struct SetLds<SET_POWERLAW, CAP> {
    ObjLds<CAP> obj;
    PowerlawLds<CAP> fit;
};

template <int CAP>
struct SetLds<SET_TDE, CAP> {
    ObjLds<CAP> obj;
    TdeLds<CAP> s;
};
template <int CAP>
struct SetLds<SET_COLOR, CAP> {
    ObjLds<CAP> obj;
    ColorLds<CAP> s;
};
template <int CAP>
struct SetLds<SET_SHAPE, CAP> {
    ObjLds<CAP> obj;
    ShapeLds<CAP> s;
};
template <int CAP>
struct SetLds<SET_PHYSICS, CAP> {
    ObjLds<CAP> obj;
    PhysicsLds<CAP> s;
};

template <int CAP>
struct SetLds<SET_RESEARCH, CAP> {
    ObjLds<CAP> obj;
    ResearchLds<CAP> s;
};

// copy `ncol` wave-shared doubles to the object's output row (coalesced on the device)
template <class W>
LCFE_FN void store_row(const double* src, double* row, int ncol) {
    for (int k = W::lane(); k < ncol; k += W::LANES) row[k] = src[k];
}
template <class W>
LCFE_FN void fill_row_nan(double* row, int ncol) {
    for (int k = W::lane(); k < ncol; k += W::LANES) row[k] = qnan();
}

template <class W, int SET, int CAP>
struct RunSet;

// policy of one bounded fit inside a wave: on the device the six band fits of a light curve run
// side by side in 8-lane groups; the host simulation runs them one after the other
template <class W> struct FitPolicy { using type = W; };
#if defined(__HIPCC__)
template <> struct FitPolicy<WaveDev> { using type = GroupDev<8>; };
#endif



template <class W, int CAP>
struct RunSet<W, SET_STAT, CAP> {
    static LCFE_FN int run(const ObjIn& in, SetLds<SET_STAT, CAP>& ws, double* row, int32_t*) {
        LCFE_PT0();
        stage_object<W, CAP>(in, ws.obj);
        LCFE_PT(0);
        stat_object<W, typename FitPolicy<W>::type, CAP>(ws.obj, ws.stat);
        LCFE_PT0B();
        store_row<W>(ws.stat.out, row, STAT_NCOL);
        W::sync();
        LCFE_PT(3);
        return 0;
    }
};

template <class W, int CAP>
struct RunSet<W, SET_BAZIN, CAP> {
    static LCFE_FN int run(const ObjIn& in, SetLds<SET_BAZIN, CAP>& ws, double* row, int32_t* st) {
        stage_object<W, CAP>(in, ws.obj);
        bazin_object<typename FitPolicy<W>::type, W, CAP>(ws.obj, ws.fit, st);
        store_row<W>(ws.fit.out, row, BAZIN_NCOL);
        W::sync();
        return 0;
    }
};

template <class W, int CAP>
struct RunSet<W, SET_POWERLAW, CAP> {
    static LCFE_FN int run(const ObjIn& in, SetLds<SET_POWERLAW, CAP>& ws, double* row, int32_t* st) {
        stage_object<W, CAP>(in, ws.obj);
        powerlaw_object<typename FitPolicy<W>::type, W, CAP>(ws.obj, ws.fit, st);
        store_row<W>(ws.fit.out, row, POWERLAW_NCOL);
        W::sync();
        return 0;
    }
};

template <class W, int CAP>
struct RunSet<W, SET_TDE, CAP> {
    static LCFE_FN int run(const ObjIn& in, SetLds<SET_TDE, CAP>& ws, double* row, int32_t*) {
        stage_object<W, CAP>(in, ws.obj);
        tde_object<W, CAP>(ws.obj, ws.s);
        store_row<W>(ws.s.out, row, TDE_NCOL);
        W::sync();
        return 0;
    }
};
template <class W, int CAP>
struct RunSet<W, SET_COLOR, CAP> {
    static LCFE_FN int run(const ObjIn& in, SetLds<SET_COLOR, CAP>& ws, double* row, int32_t*) {
        stage_object<W, CAP>(in, ws.obj);
        color_object<W, CAP>(ws.obj, ws.s);
        store_row<W>(ws.s.out, row, COLOR_NCOL);
        W::sync();
        return 0;
    }
};
template <class W, int CAP>
struct RunSet<W, SET_SHAPE, CAP> {
    static LCFE_FN int run(const ObjIn& in, SetLds<SET_SHAPE, CAP>& ws, double* row, int32_t*) {
        stage_object<W, CAP>(in, ws.obj);
        shape_object<W, typename FitPolicy<W>::type, CAP>(ws.obj, ws.s);
        store_row<W>(ws.s.out, row, SHAPE_NCOL);
        W::sync();
        return 0;
    }
};
template <class W, int CAP>
struct RunSet<W, SET_PHYSICS, CAP> {
    static LCFE_FN int run(const ObjIn& in, SetLds<SET_PHYSICS, CAP>& ws, double* row, int32_t*) {
        stage_object<W, CAP>(in, ws.obj);
        physics_object<W, CAP>(ws.obj, in.z, ws.s);
        store_row<W>(ws.s.out, row, PHYSICS_NCOL);
        W::sync();
        return 0;
    }
};

template <class W, int CAP>
struct RunSet<W, SET_RESEARCH, CAP> {
    static LCFE_FN int run(const ObjIn& in, SetLds<SET_RESEARCH, CAP>& ws, double* row, int32_t* st) {
        stage_object<W, CAP>(in, ws.obj);
        const int rc = research_object<W, CAP>(ws.obj, in.z, ws.s);
        if (st && W::lane() == 0) st[0] = rc;
        store_row<W>(ws.s.out, row, RESEARCH_NCOL);
        W::sync();
        return rc;
    }
};

}  // namespace lcfe
