!==============================================================================
! md_run_outputs -- end-of-run files of a production run, in the reference's formats:
!   <dir>/corr_<obs>.dat, <dir>/corrmean_<obs>.dat   md_simulation_program.f90:419-526, :594-634
!   <dir>/md_final_results.txt (appended block)      md_simulation_program.f90:531-560
! <obs> = epot, ekin, etot, temp, press.  Used by md_simulation_gpu (samples from the GPU) and
! by md_stats_replay (samples from a file; CPU-only test of this module and md_stats).
!==============================================================================
module md_run_outputs
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params
  use md_stats
  implicit none
  private
  public :: write_run_statistics

contains

  subroutine write_run_statistics(dir, params, total_steps, output_interval, warmup_steps, st)
    character(len=*), intent(in) :: dir
    type(sim_params), intent(in) :: params
    integer(kind=int_kind), intent(in) :: total_steps, output_interval, warmup_steps
    type(run_statistics), intent(in) :: st

    real(kind=dp_kind), allocatable :: c(:), cn(:)
    real(kind=dp_kind) :: mean(5), std(5)
    type(thermo_coefficients) :: tc
    integer(kind=int_kind) :: lag_max, n_blocks, k
    integer :: iu, ios

    if (st%n_samples <= 0) stop 'md_simulation: no samples were taken (check warmup_steps/output_interval).'

    ! the reference evaluates the coefficients (and may stop there) before it writes the curves
    call stats_thermo(st, tc)

    lag_max = stats_lag_limit(st%n_samples)
    if (lag_max >= 0) then
      if (size(st%series, 1) < st%n_samples) stop 'md_corr_add_sample(): series buffer is full.'
      allocate(c(0:lag_max), cn(0:lag_max))
      do k = 1, N_OBS
        call autocovariance(st%series(1:st%n_samples, k), lag_max, c)
        call normalise_by_lag0(lag_max, c, cn)
        call write_curve(dir // '/corr_' // trim(OBS_TAG(k)) // '.dat', '# lag   C(lag)   C_norm(lag)', &
                         'write_corr_file(): cannot open output file.', lag_max, c, cn)
      end do
      ! at most 5 blocks, each at least lag_max + 1 samples long (:475-478)
      n_blocks = min(5, st%n_samples / (lag_max + 1))
      if (n_blocks >= 1) then
        do k = 1, N_OBS
          call block_mean_autocovariance(st%series(1:st%n_samples, k), n_blocks, lag_max, c, cn)
          call write_curve(dir // '/corrmean_' // trim(OBS_TAG(k)) // '.dat', &
                           '# lag   <C(lag)>_blocks   <C_norm(lag)>_blocks', &
                           'write_corrmean_file(): cannot open output file.', lag_max, c, cn)
        end do
      end if
      deallocate(c, cn)
    end if

    do k = 1, 5            ! Q_U, Q_K, Q_E, Q_T, Q_P are quantities 1..5
      call stats_mean_std(st, k, mean(k), std(k))
    end do

    open(newunit=iu, file=dir // '/md_final_results.txt', access='append', action='write', iostat=ios)
    if (ios /= 0) stop 'md_simulation: cannot open outputs/one_run/md_final_results.txt'
    write(iu, '(a)') '************** MD PRODUCTION RESULTS **************'
    write(iu, '(a,1x,i8)')       'num_particles:', params%n
    write(iu, '(a,1x,i8)')       'num_cells:', params%num_cells
    write(iu, '(a,1x,1pe19.12)') 'box_length:', params%box_length
    write(iu, '(a,1x,1pe19.12)') 'volume:', params%volume
    write(iu, '(a,1x,1pe19.12)') 'density:', dble(params%n) / params%volume
    write(iu, '(a,1x,1pe19.12)') 'time_step:', params%dt
    write(iu, '(a,1x,i8)')       'output_interval:', output_interval
    write(iu, '(a,1x,i10)')      'total_steps:', total_steps
    write(iu, '(a,1x,i10)')      'warmup_steps:', warmup_steps
    write(iu, '(a)') '-------------------- Averages --------------------'
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') '<Epot>:', mean(Q_U), 'std:', std(Q_U)
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') '<Ekin>:', mean(Q_K), 'std:', std(Q_K)
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') '<Etot>:', mean(Q_E), 'std:', std(Q_E)
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') '<T>   :', mean(Q_T), 'std:', std(Q_T)
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') '<P>   :', mean(Q_P), 'std:', std(Q_P)
    write(iu, '(a)') '-------------- Thermodynamic coefficients --------------'
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') 'Temperature:', tc%temperature, 'Pressure:', tc%pressure
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') 'Ca_v:', tc%Ca_v, 'Ce_v:', tc%Ce_v
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') 'Ca_p:', tc%Ca_p, 'Ce_p:', tc%Ce_p
    ! three items on a two-item format: format reversion puts Gamma on its own line, as in the reference (:555)
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') 'kappa_S:', tc%K_S_inv, 'kappa_T:', tc%K_T_inv, 'Gamma:', tc%gamma
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') 'Alpha_E1:', tc%alpha_E1, 'Alpha_E2:', tc%alpha_E2
    write(iu, '(a,1x,1pe19.12,2x,a,1x,1pe19.12)') 'Alpha_S:', tc%alpha_S, 'Alpha_P:', tc%alpha_P
    write(iu, '(a)') '--------------------------------------------------------'
    write(iu, *)
    close(iu)
  end subroutine write_run_statistics

  subroutine write_curve(filename, header, errmsg, lag_max, c, cn)
    character(len=*), intent(in) :: filename, header, errmsg
    integer(kind=int_kind), intent(in) :: lag_max
    real(kind=dp_kind), intent(in) :: c(0:), cn(0:)
    integer :: iu, ios
    integer(kind=int_kind) :: lag
    open(newunit=iu, file=filename, status='replace', action='write', iostat=ios)
    if (ios /= 0) then
      write(*, '(a)') errmsg
      stop 1
    end if
    write(iu, '(a)') header
    do lag = 0, lag_max
      write(iu, '(i8,2(2x,1pe19.12))') lag, c(lag), cn(lag)
    end do
    close(iu)
  end subroutine write_curve

end module md_run_outputs
