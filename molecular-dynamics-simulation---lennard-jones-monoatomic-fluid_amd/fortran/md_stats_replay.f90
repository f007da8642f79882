!==============================================================================
! md_stats_replay -- CPU-only companion of md_simulation_gpu: replays already sampled scalars
! through md_stats / md_run_outputs and writes the end-of-run files.
!   in : inputs/input_simulation_parameters.txt
!        outputs/one_run/samples.bin   stream of fp64 quadruples (epot, ekin, d_epot, dd_epot),
!                                      one per sampling instant, in sampling order
!   out: outputs/one_run/{corr_*.dat, corrmean_*.dat, md_final_results.txt}
! Exists so that the statistics host code is testable without a GPU (tests/test_stats.py feeds
! it the reference's own raw samples and compares the files with the reference's).
!==============================================================================
program md_stats_replay
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params
  use read_input_files, only: read_simulation_parameters
  use md_stats
  use md_run_outputs,   only: write_run_statistics
  implicit none

  type(sim_params) :: params
  type(run_statistics) :: st
  integer(kind=int_kind) :: total_steps, output_interval, warmup_steps, n_expected
  real(kind=dp_kind) :: rc_over_L, target_total_energy, q(4), t_inst, p_inst
  integer :: iu, ios

  call read_simulation_parameters('inputs/input_simulation_parameters.txt', params, total_steps, &
                                  output_interval, warmup_steps, rc_over_L, target_total_energy)
  n_expected = max(0, total_steps / output_interval - warmup_steps / output_interval)
  call stats_begin(st, params%n, params%volume, n_expected)

  open(newunit=iu, file='outputs/one_run/samples.bin', access='stream', form='unformatted', &
       status='old', action='read', iostat=ios)
  if (ios /= 0) stop 'md_stats_replay: cannot open outputs/one_run/samples.bin'
  do
    read(iu, iostat=ios) q
    if (ios /= 0) exit
    call stats_push(st, q(1), q(2), q(3), q(4), t_inst, p_inst)
  end do
  close(iu)

  call write_run_statistics('outputs/one_run', params, total_steps, output_interval, warmup_steps, st)
end program md_stats_replay
