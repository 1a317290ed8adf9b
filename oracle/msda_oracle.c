/*
 * TEST INFRASTRUCTURE ONLY.  CPU oracle for the MSDeformAttn hot path: a plain-C restatement of
 * the reference's forward/backward arithmetic (src/models/ops/src/cuda/ms_deform_im2col_cuda.cuh
 * and ms_deform_attn_cuda.cu), used by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the CHECKER.  The product (devis_amd/) never links, imports or calls it.
 *
 * Pinning: checked in tests/test_oracle.py against golden vectors produced by the reference's own
 * Python oracle ms_deform_attn_core_pytorch (+ autograd) -- see tests/golden/make_golden.py.
 */
#include <math.h>
#include <stdint.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FLOOR floorf
#define FN(name) CAT(name, _f32)
#include "msda_oracle_body.inc"
#undef REAL
#undef FLOOR
#undef FN

#define REAL double
#define FLOOR floor
#define FN(name) CAT(name, _f64)
#include "msda_oracle_body.inc"
#undef REAL
#undef FLOOR
#undef FN

int msda_oracle_version(void) { return 1; }
