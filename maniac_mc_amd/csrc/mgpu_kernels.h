// HIP kernels of the GCMC energy engine, written for gfx950 (CDNA4, wave64) only, in four parts:
//   mgpu_kernels_common.h    constants, device views of topology and box, data layout in HBM, folds, Coulomb table, pair term
//   mgpu_kernels_pair.h      pair sweeps (plane by plane, flat, frozen)
//   mgpu_kernels_recip.h     reciprocal-space update (per k, rows, wide rows on the matrix units), trial geometry, S(k), intra sum
//   mgpu_kernels_windows.h   one launch per window: single chain (K steps) and farm (one lock step of many chains)
#ifndef MGPU_KERNELS_H
#define MGPU_KERNELS_H

#include "mgpu_kernels_common.h"
#include "mgpu_kernels_pair.h"
#include "mgpu_kernels_recip.h"
#include "mgpu_kernels_windows.h"

#endif
