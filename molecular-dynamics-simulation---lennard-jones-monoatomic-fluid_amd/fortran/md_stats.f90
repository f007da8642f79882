!==============================================================================
! md_stats -- scalar statistics of a production run (SURVEY 8(f) #4), host side.
!
! One table-driven module instead of the reference's four (stats_math, md_means,
! md_correlations, thermodynamic_coefs); the numbers it produces are the reference's:
!   * running means / standard deviations of 11 per-sample quantities
!       scripts/stats/md_means.f90:215-270 (sample), :311-364 (mean, population std)
!   * centred autocovariance C(lag) and C(lag)/C(0) of the five sampled observables
!       scripts/stats/stats_math.f90:129-149 (estimator), :160-190 (normalisation)
!       scripts/stats/md_correlations.f90:336-355
!   * block-averaged autocovariance (contiguous blocks of floor(n/B) samples)
!       scripts/stats/md_correlations.f90:692-799
!   * microcanonical thermodynamic coefficients from the means
!       scripts/physics/thermodynamic_coefs.f90:104-203
! Each expression keeps the reference's operand order so that the 13-digit output files
! agree digit for digit when fed the same samples.  No device code: a few kflop per run.
!==============================================================================
module md_stats
  use define_precision, only: dp_kind, int_kind
  implicit none
  private

  ! sampled observables kept as time series (columns of run_statistics%series)
  integer, parameter, public :: N_OBS = 5
  integer, parameter, public :: OBS_EPOT = 1, OBS_EKIN = 2, OBS_ETOT = 3, OBS_TEMP = 4, OBS_PRESS = 5
  character(len=5), parameter, public :: OBS_TAG(N_OBS) = [character(len=5) :: 'epot', 'ekin', 'etot', 'temp', 'press']

  ! quantities with running first and second moments (md_means.f90:82-112)
  integer, parameter, public :: N_Q = 11
  integer, parameter, public :: Q_U = 1, Q_K = 2, Q_E = 3, Q_T = 4, Q_P = 5, Q_KINV = 6, Q_DU = 7, Q_DDU = 8, &
                                Q_DU_KINV = 9, Q_DDU_KINV = 10, Q_DU2_KINV = 11

  type, public :: run_statistics
    integer(kind=int_kind) :: n_particles = 0
    real(kind=dp_kind)     :: volume = 0.d0
    integer(kind=int_kind) :: n_samples = 0
    real(kind=dp_kind)     :: s1(N_Q) = 0.d0, s2(N_Q) = 0.d0      ! sum x, sum x^2
    real(kind=dp_kind), allocatable :: series(:, :)               ! (capacity, N_OBS)
  end type run_statistics

  type, public :: thermo_coefficients
    real(kind=dp_kind) :: temperature = 0.d0, pressure = 0.d0
    real(kind=dp_kind) :: Ca_v = 0.d0, Ce_v = 0.d0, Ca_p = 0.d0, Ce_p = 0.d0
    real(kind=dp_kind) :: gamma = 0.d0, K_S = 0.d0, K_T = 0.d0, K_S_inv = 0.d0, K_T_inv = 0.d0
    real(kind=dp_kind) :: alpha_E1 = 0.d0, alpha_E2 = 0.d0, alpha_S = 0.d0, alpha_P = 0.d0
  end type thermo_coefficients

  public :: stats_begin, stats_push, stats_mean_std, stats_lag_limit
  public :: autocovariance, normalise_by_lag0, block_mean_autocovariance, stats_thermo

contains

  ! capacity = number of samples the run will push (0 is allowed: moments only)
  subroutine stats_begin(st, n_particles, volume, capacity)
    type(run_statistics), intent(inout) :: st
    integer(kind=int_kind), intent(in) :: n_particles, capacity
    real(kind=dp_kind), intent(in) :: volume
    if (n_particles <= 0) stop 'md_means_init(): params%n must be > 0.'
    if (volume <= 0.d0)   stop 'md_means_init(): params%volume must be > 0.'
    st%n_particles = n_particles
    st%volume = volume
    st%n_samples = 0
    st%s1 = 0.d0
    st%s2 = 0.d0
    if (allocated(st%series)) deallocate(st%series)
    allocate(st%series(max(capacity, 0), N_OBS))
    st%series = 0.d0
  end subroutine stats_begin

  ! One sampling instant.  Returns T = 2K/(3N) (3N, not 3N-3: md_means.f90:221) and
  ! P = rho*T + W/(3V) with W = -d_epot (md_means.f90:227, md_simulation_program.f90:366).
  subroutine stats_push(st, epot, ekin, d_epot, dd_epot, temp_inst, press_inst)
    type(run_statistics), intent(inout) :: st
    real(kind=dp_kind), intent(in)  :: epot, ekin, d_epot, dd_epot
    real(kind=dp_kind), intent(out) :: temp_inst, press_inst
    real(kind=dp_kind) :: npd, rho, virial, kinv, q(N_Q)
    integer :: k

    npd = dble(st%n_particles)
    rho = npd / st%volume
    virial = -d_epot
    temp_inst = 2.d0 * ekin / (3.d0 * npd)
    press_inst = rho * temp_inst + virial / (3.d0 * st%volume)
    if (ekin <= 0.d0) stop 'md_means_add_sample(): ekin must be > 0 to accumulate 1/ekin terms.'
    kinv = 1.d0 / ekin

    q(Q_U) = epot;  q(Q_K) = ekin;  q(Q_E) = epot + ekin;  q(Q_T) = temp_inst;  q(Q_P) = press_inst
    q(Q_KINV) = kinv
    q(Q_DU) = d_epot
    q(Q_DDU) = dd_epot
    q(Q_DU_KINV) = d_epot * kinv
    q(Q_DDU_KINV) = dd_epot * kinv
    q(Q_DU2_KINV) = (d_epot * d_epot) * kinv
    do k = 1, N_Q
      st%s1(k) = st%s1(k) + q(k)
      st%s2(k) = st%s2(k) + q(k) * q(k)
    end do

    st%n_samples = st%n_samples + 1
    if (st%n_samples <= size(st%series, 1)) then
      st%series(st%n_samples, OBS_EPOT)  = epot
      st%series(st%n_samples, OBS_EKIN)  = ekin
      st%series(st%n_samples, OBS_ETOT)  = epot + ekin
      st%series(st%n_samples, OBS_TEMP)  = temp_inst
      st%series(st%n_samples, OBS_PRESS) = press_inst
    end if
  end subroutine stats_push

  ! mean = sum/n (as sum * (1/n)); std = sqrt(max(0, <x^2> - <x>^2))   (md_means.f90:312-364, stats_math.f90:64)
  subroutine stats_mean_std(st, which, mean, std)
    type(run_statistics), intent(in) :: st
    integer, intent(in) :: which
    real(kind=dp_kind), intent(out) :: mean, std
    real(kind=dp_kind) :: inv_ns, m2
    if (st%n_samples <= 0) stop 'md_means_get(): no samples accumulated.'
    inv_ns = 1.d0 / dble(st%n_samples)
    mean = st%s1(which) * inv_ns
    m2 = st%s2(which) * inv_ns
    std = dsqrt(max(0.d0, m2 - mean * mean))
  end subroutine stats_mean_std

  ! largest lag of the full-series curves: min(1000, n-1, n/2); -1 = fewer than 2 samples, no
  ! correlations (md_simulation_program.f90:280-288)
  pure function stats_lag_limit(n_samples) result(lag_max)
    integer(kind=int_kind), intent(in) :: n_samples
    integer(kind=int_kind) :: lag_max
    if (n_samples < 2) then
      lag_max = -1
    else
      lag_max = min(min(1000, n_samples - 1), n_samples / 2)
    end if
  end function stats_lag_limit

  ! C(lag) = sum_k (x_k - m)(x_{k+lag} - m) / (n - lag), m = mean of the whole segment
  subroutine autocovariance(x, lag_max, c)
    real(kind=dp_kind), intent(in) :: x(:)
    integer(kind=int_kind), intent(in) :: lag_max
    real(kind=dp_kind), intent(out) :: c(0:)
    integer(kind=int_kind) :: n, lag, nv
    real(kind=dp_kind) :: m
    n = size(x)
    if (n <= 0)        stop 'autocorr_scalar_centered(): n_samples must be > 0.'
    if (lag_max < 0)   stop 'autocorr_scalar_centered(): lag_max must be >= 0.'
    if (lag_max >= n)  stop 'autocorr_scalar_centered(): lag_max must be < n_samples.'
    if (ubound(c, 1) < lag_max) stop 'autocorr_scalar_centered(): corr_out upper bound < lag_max.'
    m = sum(x(1:n)) / dble(n)
    do lag = 0, lag_max
      nv = n - lag
      c(lag) = dot_product(x(1:nv) - m, x(1 + lag:lag + nv) - m) / dble(nv)
    end do
  end subroutine autocovariance

  ! c / c(0); all zeros when |c(0)| <= 1e-14 (stats_math.f90:178-188)
  subroutine normalise_by_lag0(lag_max, c, cn)
    integer(kind=int_kind), intent(in) :: lag_max
    real(kind=dp_kind), intent(in)  :: c(0:)
    real(kind=dp_kind), intent(out) :: cn(0:)
    integer(kind=int_kind) :: lag
    if (abs(c(0)) <= 1.d-14) then
      cn(0:lag_max) = 0.d0
      return
    end if
    do lag = 0, lag_max
      cn(lag) = c(lag) / c(0)
    end do
  end subroutine normalise_by_lag0

  ! mean over n_blocks contiguous blocks of the per-block (C, C/C(0)); samples past
  ! n_blocks*floor(n/n_blocks) are unused (md_correlations.f90:692-799)
  subroutine block_mean_autocovariance(x, n_blocks, lag_max, c_mean, cn_mean)
    real(kind=dp_kind), intent(in) :: x(:)
    integer(kind=int_kind), intent(in) :: n_blocks, lag_max
    real(kind=dp_kind), intent(out) :: c_mean(0:), cn_mean(0:)
    real(kind=dp_kind), allocatable :: c(:), cn(:)
    real(kind=dp_kind) :: inv_nb
    integer(kind=int_kind) :: b, first, block_len
    if (n_blocks <= 0) stop 'md_corr_cm_compute(): invalid number of blocks.'
    block_len = size(x) / n_blocks
    if (block_len <= 0)       stop 'md_corr_cm_compute(): block_len <= 0 (too many blocks).'
    if (lag_max >= block_len) stop 'md_corr_cm_compute(): max_lag must be < block_len.'
    allocate(c(0:lag_max), cn(0:lag_max))
    c_mean(0:lag_max) = 0.d0
    cn_mean(0:lag_max) = 0.d0
    do b = 1, n_blocks
      first = (b - 1) * block_len + 1
      call autocovariance(x(first:first + block_len - 1), lag_max, c)
      call normalise_by_lag0(lag_max, c, cn)
      c_mean(0:lag_max)  = c_mean(0:lag_max) + c
      cn_mean(0:lag_max) = cn_mean(0:lag_max) + cn
    end do
    inv_nb = 1.d0 / dble(n_blocks)
    c_mean(0:lag_max)  = c_mean(0:lag_max) * inv_nb
    cn_mean(0:lag_max) = cn_mean(0:lag_max) * inv_nb
  end subroutine block_mean_autocovariance

  ! thermodynamic_coefs.f90:104-203.  f = 3N - 3 here (the time series uses 3N).
  subroutine stats_thermo(st, out)
    type(run_statistics), intent(in) :: st
    type(thermo_coefficients), intent(out) :: out
    real(kind=dp_kind) :: npd, f, a1, a2, vol, denom, unused
    real(kind=dp_kind) :: k_mean, p_mean, kinv, du, ddu, du_kinv, du2_kinv, ks_aux

    call stats_mean_std(st, Q_K, k_mean, unused)
    call stats_mean_std(st, Q_P, p_mean, unused)
    call stats_mean_std(st, Q_KINV, kinv, unused)
    call stats_mean_std(st, Q_DU, du, unused)
    call stats_mean_std(st, Q_DDU, ddu, unused)
    call stats_mean_std(st, Q_DU_KINV, du_kinv, unused)
    call stats_mean_std(st, Q_DU2_KINV, du2_kinv, unused)

    vol = st%volume
    npd = dble(st%n_particles)
    f = 3.d0 * npd - 3.d0
    if (f <= 0.d0) stop 'thermodynamic_compute(): degrees_of_freedom <= 0 (check N).'
    a1 = 1.d0 - 2.d0 / f
    a2 = f / 2.d0 - 1.d0

    out%temperature = 2.d0 * k_mean / f
    out%pressure = p_mean

    denom = 1.d0 - a1 * k_mean * kinv
    if (abs(denom) < 1.d-14) stop 'thermodynamic_compute(): Ca_v denominator ~ 0 (numerical instability).'
    out%Ca_v = 1.d0 / denom
    out%Ce_v = out%Ca_v / npd
    if (abs(out%Ce_v) < 1.d-14) stop 'thermodynamic_compute(): Ce_v ~ 0 (check inputs).'

    out%gamma = 1.d0 / out%Ce_v + (a2 / 3.d0) * (du * kinv - du_kinv)

    ks_aux = ((npd * out%temperature * (1.d0 + 2.d0 * out%gamma - 1.d0 / out%Ce_v)) / vol) &
             + (ddu - 2.d0 * du) / (9.d0 * vol)
    out%K_S = ks_aux - (a2 * (du2_kinv - 2.d0 * du * du_kinv + (du * du) * kinv)) / (9.d0 * vol * vol)
    if (abs(out%K_S) < 1.d-14) stop 'thermodynamic_compute(): K_S ~ 0 (cannot invert).'
    out%K_S_inv = 1.d0 / out%K_S

    out%K_T = out%K_S - (out%temperature * out%Ca_v * (out%gamma * out%gamma)) / vol
    if (abs(out%K_T) < 1.d-14) stop 'thermodynamic_compute(): K_T ~ 0 (cannot invert / compute Cp, alpha_P).'
    out%K_T_inv = 1.d0 / out%K_T

    out%Ca_p = out%Ca_v * (out%K_S / out%K_T)
    out%Ce_p = out%Ca_p / npd

    denom = (out%pressure * vol / out%Ca_v) - (out%gamma * out%temperature)
    if (abs(denom) < 1.d-14) stop 'thermodynamic_compute(): alpha_E1 denominator ~ 0.'
    out%alpha_E1 = 1.d0 / denom

    denom = (1.d0 / 3.d0) * (a1 * k_mean * du_kinv - du)
    if (abs(denom) < 1.d-14) stop 'thermodynamic_compute(): alpha_E2 denominator ~ 0.'
    out%alpha_E2 = 1.d0 / denom

    denom = out%gamma * out%temperature
    if (abs(denom) < 1.d-14) stop 'thermodynamic_compute(): gamma*T ~ 0 (alpha_S undefined).'
    out%alpha_S = -1.d0 / denom

    out%alpha_P = (out%Ca_v * out%gamma) / vol * out%K_T_inv
  end subroutine stats_thermo

end module md_stats
