/*
 * lcfe_ingest.h -- C-ABI of liblcfe_ingest.so: light-curve CSV files -> CSR arrays, on the host.
 *
 * Replaces, for callers that feed the GPU path directly, the reference's
 *   src/utils/data_loader.py:36-62   pd.read_csv per split file + pd.concat
 *   src/features/statistical.py:155  lightcurves.groupby('object_id') -> dict of frames
 * and this repository's pandas packer (mallorn-astrophysics_amd/packing.py) with one multi-threaded
 * pass over the memory-mapped files.  SURVEY.md §8(f) rank 2: at > 30 k light curves/s per GPU the
 * pandas read (~1.5 M rows/s) is the end-to-end bottleneck.
 *
 * Contract (what the pandas path produces, bit for bit):
 *   - files are read in the order given; objects are numbered by first appearance over the
 *     concatenated rows (``pd.concat(frames)['object_id'].unique()``); the rows of an object keep
 *     file order;
 *   - the columns "object_id", "Time (MJD)", "Flux", "Flux_err", "Filter" are located by header
 *     name (any order, extra columns ignored, every file has its own header);
 *   - numbers are converted exactly as pandas' default C parser does (float_precision 'high',
 *     i.e. pandas' precise_xstrtod: at most 17 significant digits accumulated in a double, one
 *     multiplication or division by a power of ten -- NOT correctly rounded, up to 1 ulp off
 *     strtod for 17-digit inputs); empty fields and pandas' default NA strings give NaN;
 *   - Filter maps u,g,r,i,z,y -> 0..5, anything else -> 255.
 * No GPU is involved; the library has no dependency besides libstdc++/pthread.
 */
#ifndef LCFE_INGEST_H
#define LCFE_INGEST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lcfe_csv lcfe_csv; /* opaque: the parsed files */

/* Parse `n_paths` CSV files with `n_threads` worker threads (<= 0: one per hardware thread).
 * Returns NULL on error (missing file, missing column, malformed number; lcfe_ingest_last_error). */
lcfe_csv* lcfe_csv_open(const char* const* paths, int n_paths, int n_threads);
void lcfe_csv_close(lcfe_csv* h);

int64_t lcfe_csv_n_objects(const lcfe_csv* h);
int64_t lcfe_csv_n_rows(const lcfe_csv* h);
/* total length of the object-id strings (no terminators) */
int64_t lcfe_csv_id_bytes(const lcfe_csv* h);

/* Copy the CSR batch into caller buffers:
 *   offsets int64[n_objects + 1], t/flux/err float64[n_rows], band uint8[n_rows],
 *   id_offsets int64[n_objects + 1] + id_bytes char[lcfe_csv_id_bytes] (object ids, in order; may be NULL).
 * Returns 0 on success. */
int lcfe_csv_fill(const lcfe_csv* h, int64_t* offsets, double* t, double* flux, double* err, uint8_t* band,
                  int64_t* id_offsets, char* id_bytes);

/* The number conversion on its own (tests compare it with pandas): parses the `len` bytes at `s`
 * as one CSV field; returns 0 and stores the value, or 1 if the field is not a number. */
int lcfe_csv_parse_double(const char* s, size_t len, double* out);

const char* lcfe_ingest_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* LCFE_INGEST_H */
