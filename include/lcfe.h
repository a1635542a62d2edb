/*
 * lcfe.h -- C-ABI of the MI355X light-curve feature-extraction engine (liblcfe.so).
 *
 * The reference (MALLORN-astrophysics) has no FFI layer: its hot path is the Python call
 *     extract_*_features(lightcurves_df, [metadata_df], object_ids) -> DataFrame
 * (src/features/statistical.py:135, bazin_fitting.py:254, multiband_gp.py:347, tde_physics.py:377,
 *  colors.py:347, lightcurve_shape.py:335, physics_based.py:461, and the inline decline fits of
 *  scripts/train_v55_powerlaw.py:147-202).  Each of those functions groups the long frame by
 * object and loops over the objects in Python.  The entry points below replace the body of that
 * loop for ALL objects at once: the caller packs the frame into CSR arrays (one slice per object,
 * rows in file order) and receives one row of feature columns per object.  A maintainer binds
 * them with ctypes (INTEGRATION.md shows the stub); no torch types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on an infrastructure error (HIP failure,
 *     bad argument); the message is available from lcfe_last_error().  Numerical failures of a
 *     fit are NOT errors: they produce NaN columns, exactly as the reference's try/except blocks
 *     do (bazin_fitting.py:168-179, multiband_gp.py:192-193, train_v55_powerlaw.py:191-192).
 *   - all arrays are C-contiguous, caller-owned; the library keeps no pointer after return.
 *   - band codes: 0..5 = u,g,r,i,z,y ; 255 = any other Filter value (counted only by the
 *     all-band statistics, as the reference's per-band filters would skip it).
 *   - integer-valued columns (*_n_obs, peak_band) are returned as exact doubles.
 */
#ifndef LCFE_H
#define LCFE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* feature-set ids; a mask is an OR of (1 << id).  Output columns of a multi-set call are the
 * concatenation of the sets' columns in increasing id order. */
enum {
    LCFE_SET_STAT = 0,     /* statistical.py:135-226        123 columns */
    LCFE_SET_BAZIN = 1,    /* bazin_fitting.py:254-288       52 columns */
    LCFE_SET_POWERLAW = 2, /* train_v55_powerlaw.py:147-202  27 columns */
    LCFE_SET_TDE = 3,      /* tde_physics.py:355-411         25 columns */
    LCFE_SET_COLOR = 4,    /* colors.py:108-380              83 columns */
    LCFE_SET_SHAPE = 5,    /* lightcurve_shape.py:177-368    65 columns */
    LCFE_SET_PHYSICS = 6,  /* physics_based.py:292-502       32 columns */
    LCFE_SET_GP2D = 7,     /* multiband_gp.py:292-385        27 columns */
    LCFE_SET_GP1D = 8,     /* gaussian_process.py:173-310    21 columns (per-band scikit-learn GP) */
    LCFE_SET_RESEARCH = 9, /* research_features.py:533-600   40 columns (v115: log-log power law, nuclear proxy,
                              colour at peak, Mexican-hat power spectra, luminosity; reads z) */
    LCFE_NUM_SETS = 10
};
#define LCFE_MASK(id) (1 << (id))
#define LCFE_MASK_ALL ((1 << LCFE_NUM_SETS) - 1)

/* Per-call profile, filled when a non-NULL pointer is passed.  kernel_ms[s] is the HIP-event time
 * of feature set s's kernel(s) on the stream they were launched on.  The statistics set (with the
 * shared binning prologue) runs alone; the other sets run concurrently on internal side streams
 * forked from / joined into the caller's stream, so their times overlap (environment variable
 * LCFE_SERIAL=1 serialises them for profiling). */
typedef struct lcfe_stats {
    double kernel_ms[LCFE_NUM_SETS];
    double h2d_ms;          /* host-buffer entry point only */
    double d2h_ms;
    int64_t bytes_in;       /* algorithmic input bytes: 25 * n_points + 8 * (n_obj + 1) [+ 8 * n_obj for z] */
    int64_t bytes_out;      /* 8 * n_obj * ncols */
    int32_t launches[LCFE_NUM_SETS];
    int32_t reserved;
} lcfe_stats;

int lcfe_version(void);
/* number of visible HIP devices (0 if none) */
int lcfe_device_count(void);
/* message of the last failing call on this thread ("" if none) */
const char* lcfe_last_error(void);

/* number / name of the output columns of a mask (same order as the reference's DataFrame, see
 * mallorn-astrophysics_amd/columns.py); lcfe_colname returns NULL when j is out of range */
int64_t lcfe_ncols(int mask);
const char* lcfe_colname(int mask, int64_t j);
/* int32 status words per object for a mask: Bazin 6 x (status, nfev), power-law 27 x (status, nfev),
 * GP 4 (status, n_iter, n_eval, n_points), per-band GP 4 (L-BFGS-B evaluations of g, r, i, z; -100: band
 * longer than 159 valid points); 0 for the other sets */
int64_t lcfe_nstatus(int mask);

/*
 * Host-buffer entry point: what extract_*_features binds.  Copies the CSR batch to `device`
 * (-1 = current device), runs the kernels of every set in `mask`, copies results back.
 *   offsets  int64[n_obj+1], offsets[0] == 0, non-decreasing
 *   t, flux, err  float64[offsets[n_obj]]   band  uint8[offsets[n_obj]]
 *   z        float64[n_obj] redshift (used by LCFE_SET_PHYSICS and LCFE_SET_RESEARCH; NULL or NaN entries = 0,
 *            physics_based.py:348)
 *   out      float64[n_obj * lcfe_ncols(mask)] row-major
 *   status   int32[n_obj * lcfe_nstatus(mask)] or NULL
 */
int lcfe_extract(int mask, int device, int64_t n_obj, const int64_t* offsets, const double* t,
                 const double* flux, const double* err, const uint8_t* band, const double* z,
                 double* out, int32_t* status, lcfe_stats* prof);

/*
 * Device-resident entry point (bench / multi-GPU path): every pointer is a DEVICE pointer on
 * `device`, kernels are enqueued on `stream` (a hipStream_t; NULL = default stream) and the call
 * returns without synchronising unless `prof` is non-NULL (event times need the stream drained).
 *   max_len    an upper bound of the number of points of any object (selects the LDS tier;
 *              objects longer than the largest tier get NaN rows and status -100)
 *   workspace  device scratch of at least lcfe_workspace_bytes(mask, n_obj, n_points) bytes (ticket
 *              counters, per-tier index lists, GP scratch); owned by the call until its work on
 *              `stream` has completed -- concurrent calls need separate workspaces
 */
size_t lcfe_workspace_bytes(int mask, int64_t n_obj, int64_t n_points);
/* ... plus the slabs of the long-object tier for a batch whose longest light curve has max_len rows: light curves beyond
 * the LDS tiers (2048 rows; 1024 for the object-level fits and the research set; 767 for the 2-D GP) run with their
 * working set in global scratch, up to lcfe_max_points() rows / lcfe_gp2d_max_points() valid points.  A call whose
 * workspace holds only lcfe_workspace_bytes() leaves such objects NaN with status -100. */
size_t lcfe_workspace_bytes_for(int mask, int64_t n_obj, int64_t n_points, int64_t max_len);
int lcfe_extract_device(int mask, int device, void* stream, int64_t n_obj, int64_t n_points,
                        int64_t max_len, const int64_t* d_offsets, const double* d_t,
                        const double* d_flux, const double* d_err, const uint8_t* d_band,
                        const double* d_z, double* d_out, int32_t* d_status, void* d_workspace,
                        size_t workspace_bytes, lcfe_stats* prof);

/* lcfe_extract keeps its device staging buffers (one set per device, grown on demand) for later calls;
 * this releases them */
void lcfe_release_buffers(void);

/* largest number of points per object any kernel tier accepts (the long-object tier) */
int64_t lcfe_max_points(void);
/* largest number of VALID points (known band, finite flux and error, error > 0) per object the 2-D GP accepts */
int64_t lcfe_gp2d_max_points(void);
/* mask of the feature sets this build of the library implements */
int lcfe_implemented_mask(void);

#ifdef __cplusplus
}
#endif
#endif /* LCFE_H */
