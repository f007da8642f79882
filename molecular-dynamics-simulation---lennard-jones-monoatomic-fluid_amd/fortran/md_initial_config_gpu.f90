!==============================================================================
! md_initial_config_gpu -- initial-configuration driver on the MI355X engine: the second
! caller of the hot path (scripts/md_initial_config_program.f90:58-121).
!   FCC lattice (4 basis atoms per cell, cell loops ix > iy > iz)            :132-187
!   velocities uniform(-0.5, 0.5) from the reference's generator, seed -12345 :82, :196-209
!   centre-of-mass velocity removed                                           :218-234
!   velocities rescaled so that Epot + Ekin = target_total_energy            :244-266
!   warmup_steps velocity-Verlet steps                                        :113-116  (on the GPU, resident)
!   outputs/rv_init.dat: record 1 = rx ry rz, record 2 = vx vy vz             :285-286
! The O(N) set-up arithmetic runs on the host exactly as written in the reference; the two
! force evaluations and the warm-up loop are the library's.
!==============================================================================
program md_initial_config_gpu
  use, intrinsic :: iso_c_binding
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params, sim_state, init_state
  use read_input_files, only: read_simulation_parameters
  use random_numbers,   only: random_uniform
  use ljmd_c_api
  implicit none

  type(sim_params) :: params
  type(sim_state), target :: state
  integer(kind=int_kind) :: total_steps, output_interval, warmup_steps, seed, i, ix, iy, iz, idx
  real(kind=dp_kind) :: rc_over_L, target_total_energy, a, x0, y0, z0
  real(kind=dp_kind) :: epot, ekin, d_epot, dd_epot, ekin_new, scale, vcm(3)
  type(c_ptr) :: engine
  integer :: iu, ios

  call read_simulation_parameters('inputs/input_simulation_parameters.txt', params, total_steps, &
                                  output_interval, warmup_steps, rc_over_L, target_total_energy)
  call init_state(params, state)

  ! ---- FCC lattice -------------------------------------------------------------------------
  a = params%box_length / dble(params%num_cells)
  idx = 0
  do ix = 0, params%num_cells - 1
    do iy = 0, params%num_cells - 1
      do iz = 0, params%num_cells - 1
        x0 = dble(ix) * a; y0 = dble(iy) * a; z0 = dble(iz) * a
        call place(x0,             y0,             z0)
        call place(x0,             y0 + 0.5d0 * a, z0 + 0.5d0 * a)
        call place(x0 + 0.5d0 * a, y0,             z0 + 0.5d0 * a)
        call place(x0 + 0.5d0 * a, y0 + 0.5d0 * a, z0)
      end do
    end do
  end do
  if (idx /= params%n) stop 'build_fcc_lattice(): unexpected particle count.'

  ! ---- velocities ---------------------------------------------------------------------------
  seed = -12345_int_kind
  do i = 1, params%n
    state%vx(i) = random_uniform(seed) - 0.5d0
    state%vy(i) = random_uniform(seed) - 0.5d0
    state%vz(i) = random_uniform(seed) - 0.5d0
  end do
  vcm(1) = sum(state%vx) / dble(params%n)
  vcm(2) = sum(state%vy) / dble(params%n)
  vcm(3) = sum(state%vz) / dble(params%n)
  state%vx = state%vx - vcm(1); state%vy = state%vy - vcm(2); state%vz = state%vz - vcm(3)

  ! ---- energies at t = 0, rescale to the target total energy ------------------------------------
  call ljmd_check(ljmd_create(engine, params%n, params%box_length, params%dt, params%rc, LJMD_PRECISION_FP64, &
                              0_c_int32_t, 0_c_int32_t, 1_c_int32_t), c_null_ptr, 'ljmd_create')
  call upload()
  call ljmd_check(ljmd_compute_forces(engine, epot, d_epot, dd_epot), engine, 'ljmd_compute_forces')
  ekin = 0.5d0 * sum(state%vx * state%vx + state%vy * state%vy + state%vz * state%vz)
  ekin_new = target_total_energy - epot
  if (ekin_new <= 0.d0) stop 'rescale_velocities_to_target_energy(): target energy too low (zero or negative kinetic).'
  if (ekin <= 0.d0) stop 'rescale_velocities_to_target_energy(): ekin_old <= 0 (cannot rescale).'
  scale = sqrt(ekin_new / ekin)
  state%vx = state%vx * scale; state%vy = state%vy * scale; state%vz = state%vz * scale
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

  open(newunit=iu, file='outputs/rv_init.dat', form='unformatted', status='replace', action='write', iostat=ios)
  if (ios /= 0) stop 'write_rv_init(): cannot open output file.'
  write(iu) state%rx, state%ry, state%rz
  write(iu) state%vx, state%vy, state%vz
  close(iu)

contains

  subroutine place(x, y, z)
    real(kind=dp_kind), intent(in) :: x, y, z
    idx = idx + 1
    state%rx(idx) = x; state%ry(idx) = y; state%rz(idx) = z
  end subroutine place

  subroutine upload()
    call ljmd_check(ljmd_set_state(engine, c_loc(state%rx), c_loc(state%ry), c_loc(state%rz), &
                                   c_loc(state%vx), c_loc(state%vy), c_loc(state%vz)), engine, 'ljmd_set_state')
  end subroutine upload

end program md_initial_config_gpu
