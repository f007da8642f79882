!==============================================================================
! lj_potential_energy -- DROP-IN replacement of the reference module of the same name
! (scripts/physics/lj_potential_energy.f90).  Same module name, same public procedure,
! same argument list and intent, same `stop` guards; the O(N^2) pair loop runs on the
! MI355X through libljmd.so.  A maintainer swaps this file (and verlet.f90) for the
! reference's and links -lljmd: every caller compiles unchanged (INTEGRATION.md).
!==============================================================================
module lj_potential_energy
  use, intrinsic :: iso_c_binding, only: c_loc, c_null_ptr, c_int, c_int32_t
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params, sim_state
  use ljmd_c_api,       only: ljmd_compute_lj_potential_energy, ljmd_stateless_set_tail_corrections, ljmd_check
  implicit none
  private
  public :: compute_lj_potential_energy, use_tail_corrections

  ! the reference's switch (:36), a compile-time parameter there and here: flip it and recompile, as in the reference.
  ! The library adds the tail constants on the host; every call tells it which way this module was compiled
  ! (verlet.f90 and the thin drivers read the same parameter).
  logical, parameter :: use_tail_corrections = .true.

contains

  subroutine compute_lj_potential_energy(params, state, epot, d_epot, dd_epot)
    type(sim_params), intent(in)            :: params
    type(sim_state),  intent(inout), target :: state
    real(kind=dp_kind), intent(out)         :: epot, d_epot, dd_epot
    integer(c_int) :: status

    ! the reference's guards, same messages (lj_potential_energy.f90:77-82)
    if (params%n <= 0_int_kind)      stop 'compute_lj_potential_energy(): params%n must be > 0.'
    if (params%box_length <= 0.d0)   stop 'compute_lj_potential_energy(): params%box_length must be > 0.'
    if (params%volume <= 0.d0)       stop 'compute_lj_potential_energy(): params%volume must be > 0.'
    if (params%rc <= 0.d0)           stop 'compute_lj_potential_energy(): params%rc must be > 0.'
    if (params%rc_square <= 0.d0)    stop 'compute_lj_potential_energy(): params%rc_square must be > 0.'
    if (.not. allocated(state%rx))   stop 'compute_lj_potential_energy(): state arrays are not allocated.'

    call ljmd_stateless_set_tail_corrections(merge(1_c_int32_t, 0_c_int32_t, use_tail_corrections))
    status = ljmd_compute_lj_potential_energy(params%n, params%box_length, params%rc,          &
                 c_loc(state%rx), c_loc(state%ry), c_loc(state%rz),                            &
                 c_loc(state%ax), c_loc(state%ay), c_loc(state%az), epot, d_epot, dd_epot)
    call ljmd_check(status, c_null_ptr, 'compute_lj_potential_energy()')
  end subroutine compute_lj_potential_energy

end module lj_potential_energy
