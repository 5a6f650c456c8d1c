/*
 * rbd_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * CPU restatement of RigidBodyDynamics.jl's dynamics!/inverse_dynamics!/mass_matrix!/dynamics_bias!
 * (see rbd_oracle_impl.h for the per-function reference citations and the parity-pin statement).
 * Built by oracle/Makefile into oracle/librbd_oracle.so; loaded only by tests/, smoke() and
 * bench.py's cpu_baseline leg.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "rbd_hip.h"

#define REAL double
#define SFX _f64
#define SIN sin
#define COS cos
#define SQRT sqrt
#include "rbd_oracle_impl.h"
#undef REAL
#undef SFX
#undef SIN
#undef COS
#undef SQRT

#define REAL float
#define SFX _f32
#define SIN sinf
#define COS cosf
#define SQRT sqrtf
#include "rbd_oracle_impl.h"
#undef REAL
#undef SFX
#undef SIN
#undef COS
#undef SQRT
