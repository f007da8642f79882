// ljmd_multi.h -- single-process multi-device engine (ljmd_create_multi): internal entry points the
// C ABI functions of ljmd_capi.cpp dispatch to when a handle is a multi-device parent.
#ifndef LJMD_MULTI_H
#define LJMD_MULTI_H

#include "ljmd_engine.h"

namespace ljmdm {

int create(ljmd_t **out, int32_t n, double box_length, double dt, double rc, int32_t precision_mode,
           int32_t n_gpus, const int32_t *devices);
void destroy(ljmd_t *h);
int set_state(ljmd_t *h, const double *rx, const double *ry, const double *rz, const double *vx, const double *vy,
              const double *vz);
int set_accel(ljmd_t *h, const double *ax, const double *ay, const double *az);
int set_unwrapped(ljmd_t *h, const double *ux, const double *uy, const double *uz);
int get_state(ljmd_t *h, double *const p[12]);
int compute_forces(ljmd_t *h, double *epot, double *d_epot, double *dd_epot);
int enqueue_steps(ljmd_t *h, int32_t nsteps, bool sampled);
int set_observables(ljmd_t *h, bool on);
int32_t migrations(const ljmd_t *h);      // ownership migrations done so far
int migrate_now(ljmd_t *h);               // ljmd_migrate on a multi-device handle
ljmd_t *rank_engine(ljmd_t *h, int32_t rank);   // NULL when out of range
int collect_steps(ljmd_t *h, int32_t nsteps, double *epot, double *ekin, double *d_epot, double *dd_epot);
int snapshot_begin(ljmd_t *h);
int snapshot_end(ljmd_t *h, double *const p[12]);
int kinetic_energy(ljmd_t *h, double *ekin);
int synchronize(ljmd_t *h);
int profile_enable(ljmd_t *h, int32_t on);
int profile_read_ex(ljmd_t *h, double *ms_avg, double *ms_min, int32_t *launches);
const char *pair_kernel_name(const ljmd_t *h);
int32_t comm_size(const ljmd_t *h);

}  // namespace ljmdm
#endif
