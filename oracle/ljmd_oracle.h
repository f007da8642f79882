/* ljmd_oracle.h -- TEST INFRASTRUCTURE: C restatement of the reference hot path.
 * See ljmd_oracle.c for the contract and the reference line citations. */
#ifndef LJMD_ORACLE_H
#define LJMD_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mirror of type(sim_params), scripts/base/md_types.f90:27-50 (legacy fields dropped) */
typedef struct ora_params {
    int32_t n;
    int32_t num_cells;
    double box_length;
    double inv_box_length;
    double volume;
    double density;
    double dt;
    double dt_half;
    double dt_square_half;
    double rc;
    double rc_square;
} ora_params;

typedef struct ora_ran3_state {
    double ma[56]; /* 1-based like the reference */
    int32_t inext, inextp, iff;
} ora_ran3_state;

int ora_derive_params(ora_params *p, int32_t n, double box_length, double dt, double rc);
double ora_minimum_image(double dx, double box_length, double inv_box_length);
void ora_wrap_positions(double *rx, double *ry, double *rz, int32_t n, double box_length);
void ora_tail_corrections(const ora_params *p, double *t_epot, double *t_d, double *t_dd);
/* the reference's compile-time switch use_tail_corrections (lj_potential_energy.f90:36), 1 = as shipped */
void ora_set_tail_corrections(int on);
void ora_compute_lj_potential_energy(const ora_params *p,
                                     const double *rx, const double *ry, const double *rz,
                                     double *ax, double *ay, double *az,
                                     double *epot, double *d_epot, double *dd_epot);
void ora_verlet_step(const ora_params *p,
                     double *rx, double *ry, double *rz,
                     double *vx, double *vy, double *vz,
                     double *ax, double *ay, double *az,
                     double *epot, double *ekin, double *d_epot, double *dd_epot);
double ora_ekin_fused(const double *vx, const double *vy, const double *vz, int32_t n);
void ora_unwrapped_update(const ora_params *p,
                          const double *rx, const double *ry, const double *rz,
                          const double *px, const double *py, const double *pz,
                          double *ux, double *uy, double *uz);
void ora_observables(const ora_params *p, double epot, double ekin, double d_epot,
                     double *etot, double *temp, double *press);
void ora_run_steps(const ora_params *p, int32_t nsteps,
                   double *rx, double *ry, double *rz,
                   double *ux, double *uy, double *uz,
                   double *vx, double *vy, double *vz,
                   double *ax, double *ay, double *az,
                   double *scalars);
void ora_rows_raw(const ora_params *p, int32_t i0, int32_t i1,
                  const double *rx, const double *ry, const double *rz,
                  double *ax, double *ay, double *az,
                  double *s_epot, double *s_d, double *s_dd);
double ora_ran3(ora_ran3_state *st, int32_t *seed);
void ora_build_fcc_lattice(int32_t num_cells, double box_length,
                           double *rx, double *ry, double *rz);

#ifdef __cplusplus
}
#endif
#endif
