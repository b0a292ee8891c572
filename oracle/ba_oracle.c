/*
 * oracle/ba_oracle.c -- builds the two precisions of the BA restatement (see ba_oracle_impl.h).
 * TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline); never on the product path.
 * PARITY UNPINNED: see the header of ba_oracle_impl.h.
 *
 *   gcc -O2 -fopenmp -shared -fPIC ba_oracle.c -o libdroid_oracle.so -lm      (oracle/Makefile)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MIN_DEPTH 0.25 /* /root/reference/src/droid_kernels.cu:26 */

/* 1: droid_oracle_ba_* rounds dx, dz, poses and disps to float32 after every iteration (the reference's
 * tensor dtypes; see the comment in droid_oracle_ba).  Arithmetic stays in REAL. */
static int g_storage_f32 = 0;
void droid_oracle_set_storage_f32(int on) { g_storage_f32 = on; }

#define REAL double
#define SUFFIX _f64
#define SIN sin
#define COS cos
#include "ba_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef SIN
#undef COS

#define REAL float
#define SUFFIX _f32
#define SIN sinf
#define COS cosf
#include "ba_oracle_impl.h"
#undef REAL
#undef SUFFIX
#undef SIN
#undef COS

int droid_oracle_abi_version(void) { return 1; }
