/*
 * isaaclab_amd — C ABI of the MI355X (gfx950) post-physics rollout kernels.
 *
 * The reference (Isaac Lab) has no FFI for this path: its seams are Python call signatures
 * (manager term functions, ActuatorBase.compute, raycast_mesh, DifferentialIKController,
 * isaaclab.utils.math).  Each entry point below names the reference interface it stands behind
 * (paths relative to the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless the name ends in _host;
 *   - the library never allocates or frees caller buffers and never synchronises the stream;
 *     work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream);
 *   - all tensors are dense row-major float32 unless stated; quaternions are (w, x, y, z);
 *   - every function returns ILX_OK (0) or a negative error code and leaves a message retrievable
 *     with ilx_last_error() (thread-local);
 *   - functions are re-entrant; the only library-owned state is inside ilx_mesh handles, which are
 *     immutable after creation.
 */
#ifndef ISAACLAB_AMD_H
#define ISAACLAB_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ILX_OK 0
#define ILX_EINVAL (-1) /* bad argument value (null pointer, unknown op, negative size) */
#define ILX_ESHAPE (-2) /* shape outside what the kernel supports */
#define ILX_EHIP (-3)   /* HIP runtime error; message holds hipGetErrorString */

#define ILX_ABI_VERSION 1

int ilx_version(void);
const char* ilx_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * isaaclab.utils.math — source/isaaclab/isaaclab/utils/math.py
 * One launch evaluates one function over n rows.  Operand meaning per op:
 *   a, b, c, d : inputs (row widths in the table), out0/out1 : outputs.
 * ------------------------------------------------------------------------------------------------ */
enum ilx_math_op {
    ILX_QUAT_ROTATE = 0,         /* math.py:583  a=q(4) b=v(3)            -> out0(3) */
    ILX_QUAT_ROTATE_INVERSE = 1, /* math.py:606  a=q(4) b=v(3)            -> out0(3) */
    ILX_QUAT_APPLY = 2,          /* math.py:546  a=q(4) b=v(3)            -> out0(3) */
    ILX_QUAT_APPLY_YAW = 3,      /* math.py:568  a=q(4) b=v(3)            -> out0(3) */
    ILX_YAW_QUAT = 4,            /* math.py:522  a=q(4)                   -> out0(4) */
    ILX_QUAT_MUL = 5,            /* math.py:464  a=q1(4) b=q2(4)          -> out0(4) */
    ILX_QUAT_CONJUGATE = 6,      /* math.py:224  a=q(4)                   -> out0(4) */
    ILX_QUAT_INV = 7,            /* math.py:239  a=q(4)                   -> out0(4) */
    ILX_QUAT_UNIQUE = 8,         /* math.py:448  a=q(4)                   -> out0(4) */
    ILX_NORMALIZE3 = 9,          /* math.py:82   a=v(3)                   -> out0(3) */
    ILX_NORMALIZE4 = 10,         /* math.py:82   a=q(4)                   -> out0(4) */
    ILX_MATRIX_FROM_QUAT = 11,   /* math.py:144  a=q(4)                   -> out0(9) */
    ILX_QUAT_BOX_MINUS = 12,     /* math.py:503  a=q1(4) b=q2(4)          -> out0(3) */
    ILX_AXIS_ANGLE_FROM_QUAT = 13, /* math.py:646 a=q(4)                  -> out0(3) */
    ILX_QUAT_ERROR_MAGNITUDE = 14, /* math.py:678 a=q1(4) b=q2(4)         -> out0(1) */
    ILX_QUAT_FROM_ANGLE_AXIS = 15, /* math.py:629 a=angle(1) b=axis(3)    -> out0(4) */
    ILX_QUAT_FROM_EULER_XYZ = 16,  /* math.py:252 a=roll(1) b=pitch(1) c=yaw(1) -> out0(4) */
    ILX_EULER_XYZ_FROM_QUAT = 17,  /* math.py:414 a=q(4)                  -> out0(3) = (roll,pitch,yaw) */
    ILX_COMBINE_FRAME = 18,      /* math.py:750  a=t01(3) b=q01(4) c=t12(3) d=q12(4) -> out0=t02(3) out1=q02(4) */
    ILX_SUBTRACT_FRAME = 19,     /* math.py:785  a=t01(3) b=q01(4) c=t02(3) d=q02(4) -> out0=t12(3) out1=q12(4) */
    ILX_POSE_ERROR_AA = 20,      /* math.py:820  a=t01(3) b=q01(4) c=t02(3) d=q02(4) -> out0=dp(3) out1=axis-angle(3) */
    ILX_POSE_ERROR_QUAT = 21,    /* math.py:820  same inputs                         -> out0=dp(3) out1=quat(4) */
    ILX_APPLY_DELTA_POSE = 22,   /* math.py:873  a=pos(3) b=quat(4) c=delta(6)       -> out0=pos(3) out1=quat(4) */
    ILX_WRAP_TO_PI = 23,         /* math.py:96   a=angle(1)               -> out0(1) */
    ILX_SKEW = 24,               /* math.py:693  a=v(3)                   -> out0(9) */
    ILX_MATH_OP_COUNT = 25
};
int ilx_math_op(int op, const float* a, const float* b, const float* c, const float* d, float* out0,
                float* out1, int64_t n, void* stream);

/* math.py:23 / :44 / :65 — scale_transform (mode 0), unscale_transform (1), saturate (2) over n x w
 * elements; lower/upper are (w,) when bounds_per_row == 0, else (n, w). */
int ilx_math_range_op(int mode, const float* x, const float* lower, const float* upper, int bounds_per_row,
                      float* out, int64_t n, int w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ISAACLAB_AMD_H */
