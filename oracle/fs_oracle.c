/*
 * fs_oracle.c - CPU oracle for the FluidSimulator.step() hot path: TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's Taichi kernels (the fs/ package of takah29/2d-fluid-simulator);
 * every function cites the reference lines it follows (see fs_oracle_impl.h).  Pinned against golden
 * vectors produced by running the reference's own kernel source under oracle/shim (tests/golden/).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fno-fast-math [-fopenmp]).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define REAL float
#define SUF f32
#define FMAX fmaxf
#define FMIN fminf
#define FABS fabsf
#define SQRT sqrtf
#include "fs_oracle_impl.h"
#undef REAL
#undef SUF
#undef FMAX
#undef FMIN
#undef FABS
#undef SQRT

#define REAL double
#define SUF f64
#define FMAX fmax
#define FMIN fmin
#define FABS fabs
#define SQRT sqrt
#include "fs_oracle_impl.h"

int oracle_abi_version(void) { return 1; }
