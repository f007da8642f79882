!==============================================================================
! md_types -- parameter and state containers of the thin driver.
!
! Component names and derived-constant expressions follow the reference's data contract
! (scripts/base/md_types.f90:27-60,132-169) so that the drop-in modules lj_potential_energy /
! verlet of this directory compile against either this module or the reference's own.
! Only what the hot path touches is kept (no inst_obs / accum_means legacy types).
!==============================================================================
module md_types
  use define_precision, only: dp_kind, int_kind
  implicit none
  private
  public :: sim_params, sim_state, pi
  public :: init_params, compute_derived_params, init_state, allocate_state, deallocate_state, zero_state

  real(kind=dp_kind), parameter :: pi = 3.1415926535897932384626433832795d0

  type :: sim_params
    integer(kind=int_kind) :: n = 0, num_cells = 0
    real(kind=dp_kind) :: box_length = 0.d0, inv_box_length = 0.d0, volume = 0.d0, density = 0.d0
    real(kind=dp_kind) :: dt = 0.d0, dt_half = 0.d0, dt_square_half = 0.d0
    real(kind=dp_kind) :: rc = 0.d0, rc_square = 0.d0
  end type sim_params

  type :: sim_state
    real(kind=dp_kind), allocatable :: rx(:), ry(:), rz(:)
    real(kind=dp_kind), allocatable :: vx(:), vy(:), vz(:)
    real(kind=dp_kind), allocatable :: ax(:), ay(:), az(:)
  end type sim_state

contains

  subroutine init_params(p, n, box_length, dt, rc, num_cells)
    type(sim_params), intent(inout) :: p
    integer(kind=int_kind), intent(in) :: n
    real(kind=dp_kind), intent(in) :: box_length, dt, rc
    integer(kind=int_kind), intent(in), optional :: num_cells
    p%n = n; p%box_length = box_length; p%dt = dt; p%rc = rc
    if (present(num_cells)) p%num_cells = num_cells
    call compute_derived_params(p)
  end subroutine init_params

  ! Same expressions, same order of guards as the reference (md_types.f90:136-162): the derived
  ! constants feed the kernels, so e.g. dt^2/2 must be (0.5*dt)*dt, not 0.5*(dt*dt).
  subroutine compute_derived_params(p)
    type(sim_params), intent(inout) :: p
    if (.not. (p%box_length > 0.d0)) stop 'compute_derived_params(): box_length must be > 0.'
    p%inv_box_length = 1.d0 / p%box_length
    p%volume = p%box_length**3
    if (p%n > 0) p%density = p%n / p%volume
    if (.not. (p%rc > 0.d0)) stop 'compute_derived_params(): rc (cutoff_radius) must be > 0.'
    p%rc_square = p%rc * p%rc
    if (p%rc >= 0.5d0 * p%box_length) &
      stop 'compute_derived_params(): rc (cutoff_radius) must be < L/2 (minimum image convention).'
    if (.not. (p%dt > 0.d0)) stop 'compute_derived_params(): dt must be > 0.'
    p%dt_half = 0.5d0 * p%dt
    p%dt_square_half = p%dt_half * p%dt
  end subroutine compute_derived_params

  subroutine init_state(p, s)
    type(sim_params), intent(in) :: p
    type(sim_state), intent(inout) :: s
    call allocate_state(p, s)
    call zero_state(s)
  end subroutine init_state

  subroutine allocate_state(p, s)
    type(sim_params), intent(in) :: p
    type(sim_state), intent(inout) :: s
    if (p%n <= 0) stop 'allocate_state(): params%n must be > 0.'
    call deallocate_state(s)
    allocate(s%rx(p%n), s%ry(p%n), s%rz(p%n), s%vx(p%n), s%vy(p%n), s%vz(p%n), &
             s%ax(p%n), s%ay(p%n), s%az(p%n))
  end subroutine allocate_state

  subroutine deallocate_state(s)
    type(sim_state), intent(inout) :: s
    if (allocated(s%rx)) deallocate(s%rx, s%ry, s%rz, s%vx, s%vy, s%vz, s%ax, s%ay, s%az)
  end subroutine deallocate_state

  subroutine zero_state(s)
    type(sim_state), intent(inout) :: s
    if (.not. allocated(s%rx)) return
    s%rx = 0.d0; s%ry = 0.d0; s%rz = 0.d0
    s%vx = 0.d0; s%vy = 0.d0; s%vz = 0.d0
    s%ax = 0.d0; s%ay = 0.d0; s%az = 0.d0
  end subroutine zero_state

end module md_types
