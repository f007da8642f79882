!==============================================================================
! verlet -- DROP-IN replacement of the reference module scripts/physics/verlet.f90.
! verlet_step keeps its signature; the whole step (drift, wrap, half-kick, forces,
! half-kick, kinetic energy: verlet.f90:58-95) runs on the GPU.  This strict form moves
! the nine state arrays to the device and back on every call, because an unmodified
! caller reads state%rx.. between steps (md_simulation_program.f90:303-353); the
! resident form (ljmd_verlet_steps) is what md_simulation_gpu.f90 uses.
!==============================================================================
module verlet
  use, intrinsic :: iso_c_binding, only: c_loc, c_null_ptr, c_int, c_int32_t
  use define_precision, only: dp_kind, int_kind
  use md_types,         only: sim_params, sim_state
  use ljmd_c_api,       only: ljmd_verlet_step, ljmd_stateless_set_tail_corrections, ljmd_check
  use lj_potential_energy, only: use_tail_corrections     ! the force routine verlet_step calls (verlet.f90:80) and its switch
  implicit none
  private
  public :: verlet_step

contains

  subroutine verlet_step(params, state, epot, ekin, d_epot, dd_epot)
    type(sim_params), intent(in)            :: params
    type(sim_state),  intent(inout), target :: state
    real(kind=dp_kind), intent(out)         :: epot, ekin, d_epot, dd_epot
    integer(c_int) :: status

    if (params%n <= 0_int_kind)    stop 'verlet_step(): params%n must be > 0.'             ! verlet.f90:51
    if (.not. allocated(state%rx)) stop 'verlet_step(): state arrays are not allocated.'  ! verlet.f90:52

    call ljmd_stateless_set_tail_corrections(merge(1_c_int32_t, 0_c_int32_t, use_tail_corrections))
    status = ljmd_verlet_step(params%n, params%box_length, params%dt, params%rc,               &
                 c_loc(state%rx), c_loc(state%ry), c_loc(state%rz),                            &
                 c_loc(state%vx), c_loc(state%vy), c_loc(state%vz),                            &
                 c_loc(state%ax), c_loc(state%ay), c_loc(state%az), epot, ekin, d_epot, dd_epot)
    call ljmd_check(status, c_null_ptr, 'verlet_step()')
  end subroutine verlet_step

end module verlet
