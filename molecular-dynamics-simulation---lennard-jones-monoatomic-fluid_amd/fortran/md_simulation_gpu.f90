!==============================================================================
! md_simulation_gpu -- thin Fortran driver of the MI355X engine.
!
! Same inputs and outputs as the reference's production program for the part that touches
! the hot path (scripts/md_simulation_program.f90:208-391):
!   in : inputs/input_simulation_parameters.txt, outputs/rv_init.dat
!   out: outputs/one_run/instantaneous_energies.dat   header :294, rows :374
!        outputs/one_run/rva.dat                      header :254-257, records :384-387
!        outputs/one_run/corr_*.dat, corrmean_*.dat, md_final_results.txt   :401-560
!        (module md_stats / md_run_outputs: same estimators, same formats)
! Differences: the state lives in HBM for the whole run (ljmd_verlet_steps advances up to the
! next sampling step without touching the host; r, ru, v, a come back only when a snapshot is
! written) and the per-step unwrapped-coordinate update (:339-353) happens inside the drift
! kernel; snapshots leave the GPU through ljmd_snapshot_begin/end while the next steps already run.
! Environment: LJMD_DEVICE (default 0), LJMD_ASYNC_IO (default 1), LJMD_GPUS (default 1; > 1: this one process
! drives that many devices -- ljmd_create_multi, particles sharded by index range, RCCL all-gather of positions and
! reduce-scatter of forces per step inside the library -- e.g. BASELINE config 4 with LJMD_GPUS=8; LJMD_DEVICES, a
! comma-separated device list of that length, overrides 0..LJMD_GPUS-1).
!==============================================================================
program md_simulation_gpu
  use, intrinsic :: iso_c_binding
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params, sim_state, init_state
  use read_input_files, only: read_simulation_parameters
  use ljmd_c_api
  use lj_potential_energy, only: use_tail_corrections     ! the reference's compile-time switch (lj_potential_energy.f90:36)
  use md_stats,         only: run_statistics, stats_begin, stats_push
  use md_run_outputs,   only: write_run_statistics
  implicit none

  type(sim_params) :: params
  type(sim_state), target :: state
  real(kind=dp_kind), allocatable, target :: rux(:), ruy(:), ruz(:)
  real(kind=dp_kind), allocatable, target :: s_epot(:), s_ekin(:), s_depot(:), s_ddepot(:)
  integer(kind=int_kind) :: total_steps, output_interval, warmup_steps, n_snapshots_expected
  real(kind=dp_kind) :: rc_over_L, target_total_energy
  real(kind=dp_kind) :: epot, ekin, etot, d_epot, dd_epot, time, temp_inst, press_inst, npd
  integer(kind=int_kind) :: step, count, k, num_samples
  logical :: sample_now, async_io, sampled_steps
  integer :: iu_rva, iu_out, ios, device, n_gpus
  integer(c_int32_t), allocatable, target :: device_list(:)
  character(len=256) :: env_list
  integer(kind=8) :: c0, c1, crate
  type(c_ptr) :: engine
  type(run_statistics) :: stats
  character(len=32) :: env

  call read_simulation_parameters('inputs/input_simulation_parameters.txt', params, total_steps, &
                                  output_interval, warmup_steps, rc_over_L, target_total_energy)
  call init_state(params, state)
  call read_rv_init('outputs/rv_init.dat')
  allocate(rux(params%n), ruy(params%n), ruz(params%n))

  device = 0
  call get_environment_variable('LJMD_DEVICE', env, status=ios)
  if (ios == 0 .and. len_trim(env) > 0) read(env, *) device
  async_io = .true.
  call get_environment_variable('LJMD_ASYNC_IO', env, status=ios)
  if (ios == 0 .and. len_trim(env) > 0) async_io = trim(env) /= '0'
  sampled_steps = .true.
  call get_environment_variable('LJMD_SAMPLED_STEPS', env, status=ios)
  if (ios == 0 .and. len_trim(env) > 0) sampled_steps = trim(env) /= '0'

  n_gpus = 1
  call get_environment_variable('LJMD_GPUS', env, status=ios)
  if (ios == 0 .and. len_trim(env) > 0) read(env, *) n_gpus
  if (n_gpus > 1) then
    allocate(device_list(n_gpus))
    device_list = [(int(k - 1, c_int32_t), k = 1, n_gpus)]
    call get_environment_variable('LJMD_DEVICES', env_list, status=ios)
    if (ios == 0 .and. len_trim(env_list) > 0) read(env_list, *) device_list
    call ljmd_check(ljmd_create_multi(engine, params%n, params%box_length, params%dt, params%rc, &
                                      LJMD_PRECISION_FP64, int(n_gpus, c_int32_t), c_loc(device_list)), &
                    c_null_ptr, 'ljmd_create_multi')
  else
    call ljmd_check(ljmd_create(engine, params%n, params%box_length, params%dt, params%rc, &
                                LJMD_PRECISION_FP64, int(device, c_int32_t), 0_c_int32_t, 1_c_int32_t), &
                    c_null_ptr, 'ljmd_create')
  end if
  call ljmd_check(ljmd_set_tail_corrections(engine, merge(1_c_int32_t, 0_c_int32_t, use_tail_corrections)), engine, &
                  'ljmd_set_tail_corrections')
  ! H2D; the library sets ru <- r (md_simulation_program.f90:229-231)
  call ljmd_check(ljmd_set_state(engine, c_loc(state%rx), c_loc(state%ry), c_loc(state%rz), &
                                 c_loc(state%vx), c_loc(state%vy), c_loc(state%vz)), engine, 'ljmd_set_state')
  ! t = 0 forces and energies (:236-243)
  call ljmd_check(ljmd_compute_forces(engine, epot, d_epot, dd_epot), engine, 'ljmd_compute_forces')
  call ljmd_check(ljmd_kinetic_energy(engine, ekin), engine, 'ljmd_kinetic_energy')
  etot = epot + ekin
  time = 0.d0

  open(newunit=iu_rva, file='outputs/one_run/rva.dat', form='unformatted', status='replace', &
       action='write', iostat=ios)
  if (ios /= 0) stop 'md_simulation: cannot open outputs/one_run/rva.dat'
  n_snapshots_expected = (total_steps / output_interval) - (warmup_steps / output_interval)
  if (n_snapshots_expected < 0) n_snapshots_expected = 0
  write(iu_rva) params%n, params%box_length, params%dt, output_interval, n_snapshots_expected

  open(newunit=iu_out, file='outputs/one_run/instantaneous_energies.dat', status='replace', &
       action='write', iostat=ios)
  if (ios /= 0) stop 'md_simulation: cannot open outputs/one_run/instantaneous_energies.dat'
  write(iu_out, '(a)') '# time   epot   ekin   etot   T   P'

  allocate(s_epot(LJMD_MAX_PENDING_STEPS), s_ekin(LJMD_MAX_PENDING_STEPS), &
           s_depot(LJMD_MAX_PENDING_STEPS), s_ddepot(LJMD_MAX_PENDING_STEPS))
  npd = dble(params%n)
  call stats_begin(stats, params%n, params%volume, n_snapshots_expected)
  num_samples = 0
  step = 0
  call system_clock(c0, crate)
  ! Software pipeline: while the host formats and writes the sample taken at step s, the GPU is
  ! already running the steps up to the next sampling instant.  LJMD_ASYNC_IO=0 serialises the two
  ! (the next segment is enqueued only after the files are written) for A/B timing.
  count = segment_length(step)
  call enqueue_segment(count)
  do while (step < total_steps)
    call ljmd_check(ljmd_collect_steps(engine, count, c_loc(s_epot), c_loc(s_ekin), c_loc(s_depot), &
                                       c_loc(s_ddepot)), engine, 'ljmd_collect_steps')
    do k = 1, count
      time = time + params%dt                       ! accumulated as at :356
    end do
    step = step + count
    epot = s_epot(count); ekin = s_ekin(count); d_epot = s_depot(count); dd_epot = s_ddepot(count)
    etot = epot + ekin
    sample_now = step > warmup_steps .and. mod(step, output_interval) == 0       ! :361
    if (sample_now) call ljmd_check(ljmd_snapshot_begin(engine), engine, 'ljmd_snapshot_begin')
    count = segment_length(step)
    if (async_io .and. count > 0) call enqueue_segment(count)
    if (sample_now) then
      num_samples = num_samples + 1
      call stats_push(stats, epot, ekin, d_epot, dd_epot, temp_inst, press_inst)   ! T, P as md_means.f90:221,227
      write(iu_out, '(1pe13.6,5(2x,1pe13.6))') time, epot, ekin, etot, temp_inst, press_inst
      call ljmd_check(ljmd_snapshot_end(engine, c_loc(state%rx), c_loc(state%ry), c_loc(state%rz), &
                                        c_loc(rux), c_loc(ruy), c_loc(ruz), &
                                        c_loc(state%vx), c_loc(state%vy), c_loc(state%vz), &
                                        c_loc(state%ax), c_loc(state%ay), c_loc(state%az)), engine, 'ljmd_snapshot_end')
      write(iu_rva) state%rx, state%ry, state%rz
      write(iu_rva) rux, ruy, ruz
      write(iu_rva) state%vx, state%vy, state%vz
      write(iu_rva) state%ax, state%ay, state%az
    end if
    if (.not. async_io .and. count > 0) call enqueue_segment(count)
  end do
  call system_clock(c1)
  close(iu_out)
  close(iu_rva)
  call ljmd_destroy(engine)

  if (num_samples <= 0) stop 'md_simulation: no samples were taken (check warmup_steps/output_interval).'
  call write_run_statistics('outputs/one_run', params, total_steps, output_interval, warmup_steps, stats)
  write(*, '(a,i0,a,i0,a,f10.2,a,es11.4,a)') 'md_simulation_gpu: N=', params%n, ' steps=', total_steps, &
    '  ', dble(total_steps) * dble(crate) / dble(max(c1 - c0, 1_8)), ' steps/s  ', &
    0.5d0 * npd * (npd - 1.d0) * dble(total_steps) * dble(crate) / dble(max(c1 - c0, 1_8)), ' pair-interactions/s'

contains

  ! Only the last step of a segment is read below (epot = s_epot(count) ...): the reference samples at :361 and
  ! nowhere else, so the steps in between run the forces-only pair kernel.  LJMD_SAMPLED_STEPS=0 evaluates the
  ! energy sums on every step as lj_potential_energy.f90 does (A/B timing; r, v, a are bit-identical either way).
  subroutine enqueue_segment(n_steps)
    integer(kind=int_kind), intent(in) :: n_steps
    if (sampled_steps) then
      call ljmd_check(ljmd_enqueue_steps_sampled(engine, n_steps), engine, 'ljmd_enqueue_steps_sampled')
    else
      call ljmd_check(ljmd_enqueue_steps(engine, n_steps), engine, 'ljmd_enqueue_steps')
    end if
  end subroutine

  ! steps from `from_step` to the next sampling instant of :361 (or to the end of the run), capped by
  ! the engine's pending-step limit; 0 when the run is complete
  function segment_length(from_step) result(n)
    integer(kind=int_kind), intent(in) :: from_step
    integer(kind=int_kind) :: n, nxt
    nxt = (from_step / output_interval + 1) * output_interval
    do while (nxt <= warmup_steps)
      nxt = nxt + output_interval
    end do
    nxt = min(nxt, total_steps)
    n = min(max(nxt - from_step, 0), LJMD_MAX_PENDING_STEPS)
  end function segment_length

  ! outputs/rv_init.dat: record 1 = rx ry rz, record 2 = vx vy vz (md_initial_config_program.f90:285-286)
  subroutine read_rv_init(filename)
    character(len=*), intent(in) :: filename
    integer :: iu, ierr
    open(newunit=iu, file=filename, form='unformatted', status='old', action='read', iostat=ierr)
    if (ierr /= 0) stop 'read_rv_init(): cannot open rv_init file.'
    read(iu) state%rx, state%ry, state%rz
    read(iu) state%vx, state%vy, state%vz
    close(iu)
  end subroutine read_rv_init

end program md_simulation_gpu
