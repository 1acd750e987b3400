/*
 * vg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of visgeom's calibration hot path
 *   GenericProjectionJac::Evaluate   src/calibration/calib_cost_functions.cpp:28-117
 * and of the headers it calls (geometry, quaternion, EUCM/UCM/Mei, InterJacobian).
 * Every function in vg_oracle.c cites the reference file:line it follows.
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg -- as the CHECKER / reported CPU baseline only.  The product path
 * (visgeom_amd/, include/visgeom_amd.h) never links, imports or calls this.
 *
 * Pinning status: the reference needs Eigen3 + Ceres (absent in this image, no
 * network) so it cannot be compiled here without writing stand-in headers, which
 * this build does not do.  The oracle is pinned against the known answers that
 * the reference's own code produced at survey time (SURVEY.md Appendix C,
 * transcribed to tests/golden/survey_appendix_c.json), and cross-checked against
 * central differences and an independent float64 autograd formulation
 * (tests/test_oracle_*.py).  Anything Appendix C does not exercise is
 * "parity unpinned" -- see DESIGN.md section 3.
 */
#ifndef VG_ORACLE_H
#define VG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define VGO_MODEL_EUCM 0 /* [alpha,beta,fu,fv,u0,v0]            include/projection/eucm.h  */
#define VGO_MODEL_UCM 1  /* [xi,fu,fv,u0,v0]                    include/projection/ucm.h   */
#define VGO_MODEL_MEI 2  /* [xi,k1,k2,k3,k4,k5,fu,fv,u0,v0]     include/projection/mei.h   */

#define VGO_TRANSFORM_DIRECT 0  /* include/calibration/calib_cost_functions.h:25 */
#define VGO_TRANSFORM_INVERSE 1

#define VGO_MAX_CHAIN 5    /* src/calibration/unified_calibration.cpp:566-567 */
#define VGO_DOUBLE_BIG 1e15 /* include/std.h:71 */

/* number of intrinsics of a model (6 / 5 / 10), -1 if unknown */
int vgo_num_intrinsics(int model);

/* ---- geometry primitives, exported so tests can pin each branch ---- */
void vgo_quat_from_rotvec(const double rot[3], double q[4]);          /* quaternion.h:31-50  */
void vgo_quat_to_rotvec(const double q[4], double rot[3]);            /* quaternion.h:84-98  */
void vgo_compose(const double a[6], const double b[6], double out[6]);        /* transformation.h:80-88   */
void vgo_compose_inverse(const double a[6], const double b[6], double out[6]); /* transformation.h:101-110 */
void vgo_rotation_matrix(const double v[3], double R[9]);             /* geometry_core.h:40-76   */
void vgo_inter_omega_rot(const double v[3], double M[9]);             /* geometry_core.h:158-180 */

/* ---- camera primitives: return 1 when projected, 0 when the reference returns false ---- */
int vgo_project_point(int model, const double *intr, const double X[3], double uv[2]);
int vgo_projection_jacobian(int model, const double *intr, const double X[3], double dudx[3], double dvdx[3]);
int vgo_intrinsic_jacobian(int model, const double *intr, const double X[3], double *du, double *dv);

/*
 * 1:1 restatement of GenericProjectionJac::Evaluate (calib_cost_functions.cpp:28-117).
 *   params[0]    -> K intrinsics, params[1+l] -> [tx,ty,tz,rx,ry,rz] of chain member l
 *   residual     -> 2N doubles, [u0,v0,u1,v1,...]; failed projection -> 1e15 pair
 *   jac          -> NULL, or L+1 pointers, each NULL or row-major [2N x blocksize]
 * Returns 1 (the reference always returns true) or 0 on invalid arguments.
 */
int vgo_eval_block(int model, int L, const int *status, int N, const double *grid /*3N*/,
                   const double *obs /*2N*/, const double *const *params, double *residual,
                   double **jac);

/*
 * Batched convenience used by the CPU-baseline leg: n_blocks independent blocks that
 * share one camera and one chain shape.  Parameter pointers are given per block through
 * an index table:  member l of block b lives at  param_vec + member_base[l] + member_stride[l]*seq_index[b].
 * Outputs use the Ceres block layout, block after block:
 *   residuals [n_blocks][2N],  jac_intr [n_blocks][2N][K],  jac_member[l] [n_blocks][2N][6].
 * threads <= 1 -> serial; otherwise OpenMP over blocks.  Returns number of blocks evaluated.
 */
long vgo_eval_dataset(int model, int L, const int *status, int N, const double *grid,
                      long n_blocks, const double *obs /*[n_blocks][2N]*/,
                      const double *param_vec, long intr_offset, const long *member_base,
                      const long *member_stride, const long *seq_index,
                      double *residuals, double *jac_intr, double *const *jac_member,
                      int threads);

/*
 * Per-block Gram of the stacked row block [J_0 | J_1 | ... | J_L | r]  (2N x (P+1), P = K + 6L):
 *   gram[(P+1)*(P+1)] row-major, full symmetric matrix, accumulated in long double.
 * This is the mathematical definition the normal-equation kernels are checked against
 * ("How parity is then defined").
 */
void vgo_block_gram(int K, int L, int N, const double *residual, const double *jac_intr,
                    const double *const *jac_member, double *gram);

/* same, plain double accumulation in row order (the CPU-baseline variant that is timed) */
void vgo_block_gram_fast(int K, int L, int N, const double *residual, const double *jac_intr,
                         const double *const *jac_member, double *gram);

/* CPU-baseline leg of the normal-equation build: per-block Gram (plain double) of every block, OpenMP over
 * blocks, then a serial sum over blocks.  grams [n_blocks][W*W], sum [W*W] (may be NULL). */
long vgo_dataset_gram(int K, int L, int N, long n_blocks, const double *residuals, const double *jac_intr,
                      const double *const *jac_member, double *grams, double *sum, int threads);

/* TransformationPrior::Evaluate (calib_cost_functions.h:79-103, .cpp:214-228): residual[6] = A [R e_t; R e_r],
 * e = xi_prior^-1 o xi; jac (may be NULL) = A, row-major 6x6 (the reference's constant Jacobian). */
void vgo_transformation_prior(const double stiffness[6], const double xi_prior[6], const double xi[6],
                              double residual[6], double jac[36]);

/* OdometryPrior (calibration version): constructor (calib_cost_functions.cpp:119-167) -> zetaPrior[6], A[36]
 * (row-major), and Evaluate (:171-212) -> residual[6], J1 / J2 [36] row-major (either may be NULL). */
void vgo_odometry_prior_init(double errV, double errW, double lambda, const double xi1[6], const double xi2[6],
                             double zetaPrior[6], double A[36]);
void vgo_odometry_prior_eval(const double zetaPrior[6], const double A[36], const double xi1[6], const double xi2[6],
                             double residual[6], double J1[36], double J2[36]);

/* OdometryCost (src/calibration/odometry_cost_function.cpp): constructor (:147-197) -> zetaPrior[6], A[36]; Evaluate
 * (:202-266) with parameter blocks (xi1[6], xi2[6], intrinsics[3] = wheel radii left / right, track gauge) ->
 * residual[6], J1 / J2 [36], J3 [6 x 3], row-major, any of them NULL.  deltaQ: n wheel-increment pairs. */
void vgo_odometry_cost_init(double errV, double errW, double lambda, int n, const double *deltaQ, const double intr_prior[3],
                            double zetaPrior[6], double A[36]);
void vgo_odometry_cost_eval(const double A[36], int n, const double *deltaQ, const double xi1[6], const double xi2[6],
                            const double intr[3], double residual[6], double J1[36], double J2[36], double J3[18]);

/* ---- localization costs on the same camera models ---- */
/* CameraJacobian ctors (include/projection/jacobian.h:54-71); T23 NULL = the one-transform ctor */
void vgo_camera_jacobian_init(const double T12[6], const double *T23, double L11[9], double L12[9], double L22[9]);
/* CameraJacobian::dpdxi (:75-96) and ::dfdxi (:99-113, when grad and dfdxi are given); outputs [6] each, any may be NULL */
void vgo_camera_jacobian_eval(int model, const double *intr, int two, const double L11[9], const double L12[9],
                              const double L22[9], const double X2[3], const double *grad, double *dudxi, double *dvdxi,
                              double *dfdxi);
/* Triangulator::computeRegular (src/reconstruction/triangulator.cpp:145-259); R [9] row-major, t [3] of the transform */
void vgo_triangulate_regular(const double R[9], const double t[3], double eps, const double p[3], const double q[3],
                             double *res1, double *res2, double *jac1, double *jac2);
/* MonoReprojectCost::Evaluate (src/localization/local_cost_functions.cpp:216-278): blocks [6, 5], 10 residuals */
void vgo_mono_reproject(int model, const double *intr, const double xiBaseCam[6], const double *x1 /*[5][3]*/,
                        const double *p2 /*[5][2]*/, const double xiOdom[6], const double lengths[5], double residual[10],
                        double *jac_odom /*[10][6]*/, double *jac_len /*[10][5]*/);
/* SparseReprojectCost::Evaluate (:281-391): block [6], 2n residuals */
void vgo_sparse_reproject(int model, const double *intr, const double xiBaseCam[6], int n, const double *x1, const double *x2,
                          const double *p2, const double *size, const double xiOdom[6], double *residual, double *jac);

/* ---- geometric pose initialisation ---- */
int vgo_reconstruct_point(int model, const double *intr, const double uv[2], double X[3]); /* eucm.h:85-106, ucm.h:81-103, mei.h:90-112 */
void vgo_rotation_vector(const double R[9], double rot[3]);                               /* geometry_core.h:120-124, quaternion.h:52-59 */
/* estimateInitialGrid's 4-corner construction (unified_calibration.cpp:1066-1135); board4 / corners4 in the order UL, UR, BL, BR */
int vgo_initial_grid_pose(int model, const double *intr, const double board4[12], const double corners4[8], double xi[6]);
/* getInitTransform (unified_calibration.cpp:311-348) */
void vgo_init_transform(int n, const int *status, int init_index, const double *chain, const double xi_in[6], double out[6]);
void vgo_init_transform_range(int n, const int *status, int first_index, int last_index, const double *chain, const double xi_in[6],
                              double out[6]);

int vgo_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
