!==============================================================================
! md_initial_config_gpu -- initial-configuration driver on the MI355X engine: the second
! caller of the hot path (scripts/md_initial_config_program.f90:58-121).
!   FCC lattice, ran3 velocities (seed -12345, :82), centre-of-mass removal, rescale to the
!   target total energy: module md_init_host (host arithmetic exactly as written in the reference)
!   the two force evaluations (:91, :104) and the warm-up loop (:113-116): the library, resident
!   outputs/rv_init.dat: record 1 = rx ry rz, record 2 = vx vy vz             :285-286
! Environment: LJMD_DEVICE (default 0).
!==============================================================================
program md_initial_config_gpu
  use, intrinsic :: iso_c_binding
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params, sim_state, init_state
  use read_input_files, only: read_simulation_parameters
  use md_init_host
  use ljmd_c_api
  use lj_potential_energy, only: use_tail_corrections     ! the reference's compile-time switch (lj_potential_energy.f90:36)
  implicit none

  type(sim_params) :: params
  type(sim_state), target :: state
  integer(kind=int_kind) :: total_steps, output_interval, warmup_steps, seed
  real(kind=dp_kind) :: rc_over_L, target_total_energy
  real(kind=dp_kind) :: epot, d_epot, dd_epot
  type(c_ptr) :: engine
  integer(c_int32_t) :: device
  character(len=32) :: env
  integer :: ios

  call read_simulation_parameters('inputs/input_simulation_parameters.txt', params, total_steps, &
                                  output_interval, warmup_steps, rc_over_L, target_total_energy)
  call init_state(params, state)
  device = 0
  call get_environment_variable('LJMD_DEVICE', env, status=ios)
  if (ios == 0 .and. len_trim(env) > 0) read(env, *, iostat=ios) device

  call build_fcc_lattice(params, state)
  seed = -12345_int_kind
  call assign_random_velocities(params, state, seed)
  call remove_center_of_mass_velocity(params, state)

  ! ---- energies at t = 0, rescale to the target total energy ------------------------------------
  call ljmd_check(ljmd_create(engine, params%n, params%box_length, params%dt, params%rc, LJMD_PRECISION_FP64, &
                              device, 0_c_int32_t, 1_c_int32_t), c_null_ptr, 'ljmd_create')
  call ljmd_check(ljmd_set_tail_corrections(engine, merge(1_c_int32_t, 0_c_int32_t, use_tail_corrections)), engine, &
                  'ljmd_set_tail_corrections')
  call upload()
  call ljmd_check(ljmd_compute_forces(engine, epot, d_epot, dd_epot), engine, 'ljmd_compute_forces')
  call rescale_velocities_to_target_energy(params, state, target_total_energy, epot)
  call upload()
  call ljmd_check(ljmd_compute_forces(engine, epot, d_epot, dd_epot), engine, 'ljmd_compute_forces')

  ! ---- warm-up on the device, then the hand-off file ---------------------------------------------
  if (warmup_steps > 0) &
    call ljmd_check(ljmd_verlet_steps(engine, warmup_steps, c_null_ptr, c_null_ptr, c_null_ptr, c_null_ptr), &
                    engine, 'ljmd_verlet_steps')
  call ljmd_check(ljmd_get_state(engine, c_loc(state%rx), c_loc(state%ry), c_loc(state%rz), &
                                 c_null_ptr, c_null_ptr, c_null_ptr, &
                                 c_loc(state%vx), c_loc(state%vy), c_loc(state%vz), &
                                 c_null_ptr, c_null_ptr, c_null_ptr), engine, 'ljmd_get_state')
  call ljmd_destroy(engine)
  call write_rv_init('outputs/rv_init.dat', params, state)

contains

  subroutine upload()
    call ljmd_check(ljmd_set_state(engine, c_loc(state%rx), c_loc(state%ry), c_loc(state%rz), &
                                   c_loc(state%vx), c_loc(state%vy), c_loc(state%vz)), engine, 'ljmd_set_state')
  end subroutine upload

end program md_initial_config_gpu
