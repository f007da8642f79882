!==============================================================================
! ljmd_c_api -- ISO_C_BINDING interfaces of libljmd.so (include/ljmd.h).
!
! This is the whole "FFI" a Fortran host needs: plain pointers, sizes and scalars by
! value.  The two stateless entry points replace the reference's module procedures
!   compute_lj_potential_energy   scripts/physics/lj_potential_energy.f90:46
!   verlet_step                   scripts/physics/verlet.f90:41
! and are what the drop-in modules lj_potential_energy.f90 / verlet.f90 of this
! directory forward to.  The handle-based entry points keep the state resident in
! HBM and are used by the thin driver md_simulation_gpu.f90.
!==============================================================================
module ljmd_c_api
  use, intrinsic :: iso_c_binding
  implicit none
  private

  public :: ljmd_compute_lj_potential_energy, ljmd_verlet_step, ljmd_stateless_reset
  public :: ljmd_set_tail_corrections, ljmd_stateless_set_tail_corrections
  public :: ljmd_create, ljmd_create_multi, ljmd_destroy, ljmd_set_state, ljmd_set_accel, ljmd_set_unwrapped
  public :: ljmd_get_state, ljmd_compute_forces, ljmd_verlet_steps, ljmd_kinetic_energy
  public :: ljmd_last_error, ljmd_device_count, ljmd_profile_enable, ljmd_profile_read
  public :: ljmd_enqueue_steps, ljmd_enqueue_steps_sampled, ljmd_collect_steps, ljmd_snapshot_begin, ljmd_snapshot_end
  public :: ljmd_check, ljmd_error_text

  integer(c_int), parameter, public :: LJMD_OK = 0
  integer(c_int32_t), parameter, public :: LJMD_PRECISION_FP64 = 0
  integer(c_int32_t), parameter, public :: LJMD_MAX_PENDING_STEPS = 4096

  interface
    function ljmd_compute_lj_potential_energy(n, box_length, rc, rx, ry, rz, ax, ay, az, &
                                              epot, d_epot, dd_epot) bind(C, name="ljmd_compute_lj_potential_energy") result(status)
      import :: c_int, c_int32_t, c_double, c_ptr
      integer(c_int32_t), value :: n
      real(c_double), value :: box_length, rc
      type(c_ptr), value :: rx, ry, rz, ax, ay, az
      real(c_double), intent(out) :: epot, d_epot, dd_epot
      integer(c_int) :: status
    end function

    function ljmd_verlet_step(n, box_length, dt, rc, rx, ry, rz, vx, vy, vz, ax, ay, az, &
                              epot, ekin, d_epot, dd_epot) bind(C, name="ljmd_verlet_step") result(status)
      import :: c_int, c_int32_t, c_double, c_ptr
      integer(c_int32_t), value :: n
      real(c_double), value :: box_length, dt, rc
      type(c_ptr), value :: rx, ry, rz, vx, vy, vz, ax, ay, az
      real(c_double), intent(out) :: epot, ekin, d_epot, dd_epot
      integer(c_int) :: status
    end function

    subroutine ljmd_stateless_reset() bind(C, name="ljmd_stateless_reset")
    end subroutine

    ! the reference's use_tail_corrections (lj_potential_energy.f90:36): 0 = no tail constants in epot, d_epot, dd_epot
    subroutine ljmd_stateless_set_tail_corrections(on) bind(C, name="ljmd_stateless_set_tail_corrections")
      import :: c_int32_t
      integer(c_int32_t), value :: on
    end subroutine

    function ljmd_set_tail_corrections(handle, on) bind(C, name="ljmd_set_tail_corrections") result(status)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: on
      integer(c_int) :: status
    end function

    function ljmd_create(handle, n, box_length, dt, rc, precision_mode, device, rank, n_ranks) &
        bind(C, name="ljmd_create") result(status)
      import :: c_int, c_int32_t, c_double, c_ptr
      type(c_ptr), intent(out) :: handle
      integer(c_int32_t), value :: n, precision_mode, device, rank, n_ranks
      real(c_double), value :: box_length, dt, rc
      integer(c_int) :: status
    end function

    ! one process, n_gpus devices (devices = c_null_ptr: 0 .. n_gpus-1); same entry points afterwards
    function ljmd_create_multi(handle, n, box_length, dt, rc, precision_mode, n_gpus, devices) &
        bind(C, name="ljmd_create_multi") result(status)
      import :: c_int, c_int32_t, c_double, c_ptr
      type(c_ptr), intent(out) :: handle
      integer(c_int32_t), value :: n, precision_mode, n_gpus
      real(c_double), value :: box_length, dt, rc
      type(c_ptr), value :: devices
      integer(c_int) :: status
    end function

    subroutine ljmd_destroy(handle) bind(C, name="ljmd_destroy")
      import :: c_ptr
      type(c_ptr), value :: handle
    end subroutine

    function ljmd_set_state(handle, rx, ry, rz, vx, vy, vz) bind(C, name="ljmd_set_state") result(status)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle, rx, ry, rz, vx, vy, vz
      integer(c_int) :: status
    end function

    function ljmd_set_accel(handle, ax, ay, az) bind(C, name="ljmd_set_accel") result(status)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle, ax, ay, az
      integer(c_int) :: status
    end function

    function ljmd_set_unwrapped(handle, ux, uy, uz) bind(C, name="ljmd_set_unwrapped") result(status)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle, ux, uy, uz
      integer(c_int) :: status
    end function

    function ljmd_get_state(handle, rx, ry, rz, ux, uy, uz, vx, vy, vz, ax, ay, az) &
        bind(C, name="ljmd_get_state") result(status)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle, rx, ry, rz, ux, uy, uz, vx, vy, vz, ax, ay, az
      integer(c_int) :: status
    end function

    function ljmd_compute_forces(handle, epot, d_epot, dd_epot) bind(C, name="ljmd_compute_forces") result(status)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(out) :: epot, d_epot, dd_epot
      integer(c_int) :: status
    end function

    function ljmd_verlet_steps(handle, nsteps, epot, ekin, d_epot, dd_epot) &
        bind(C, name="ljmd_verlet_steps") result(status)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: nsteps
      type(c_ptr), value :: epot, ekin, d_epot, dd_epot      ! each c_null_ptr or real(c_double)(nsteps)
      integer(c_int) :: status
    end function

    ! asynchronous production loop: enqueue returns at once, collect waits for the engine's stream
    function ljmd_enqueue_steps(handle, nsteps) bind(C, name="ljmd_enqueue_steps") result(status)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: nsteps
      integer(c_int) :: status
    end function

    ! the same, the potential-energy sums evaluated on the LAST step only (the one the caller samples)
    function ljmd_enqueue_steps_sampled(handle, nsteps) bind(C, name="ljmd_enqueue_steps_sampled") result(status)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: nsteps
      integer(c_int) :: status
    end function

    function ljmd_collect_steps(handle, nsteps, epot, ekin, d_epot, dd_epot) &
        bind(C, name="ljmd_collect_steps") result(status)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: nsteps
      type(c_ptr), value :: epot, ekin, d_epot, dd_epot      ! each c_null_ptr or real(c_double)(nsteps)
      integer(c_int) :: status
    end function

    ! snapshot of r, ru, v, a: begin = stream-ordered freeze + transfer on a second stream (returns
    ! at once), end = wait for that transfer only and deliver the arrays
    function ljmd_snapshot_begin(handle) bind(C, name="ljmd_snapshot_begin") result(status)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle
      integer(c_int) :: status
    end function

    function ljmd_snapshot_end(handle, rx, ry, rz, ux, uy, uz, vx, vy, vz, ax, ay, az) &
        bind(C, name="ljmd_snapshot_end") result(status)
      import :: c_int, c_ptr
      type(c_ptr), value :: handle, rx, ry, rz, ux, uy, uz, vx, vy, vz, ax, ay, az
      integer(c_int) :: status
    end function

    function ljmd_kinetic_energy(handle, ekin) bind(C, name="ljmd_kinetic_energy") result(status)
      import :: c_int, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(out) :: ekin
      integer(c_int) :: status
    end function

    function ljmd_last_error(handle) bind(C, name="ljmd_last_error") result(msg)
      import :: c_ptr
      type(c_ptr), value :: handle
      type(c_ptr) :: msg
    end function

    function ljmd_device_count() bind(C, name="ljmd_device_count") result(n)
      import :: c_int32_t
      integer(c_int32_t) :: n
    end function

    function ljmd_profile_enable(handle, on) bind(C, name="ljmd_profile_enable") result(status)
      import :: c_int, c_int32_t, c_ptr
      type(c_ptr), value :: handle
      integer(c_int32_t), value :: on
      integer(c_int) :: status
    end function

    function ljmd_profile_read(handle, ms_avg, launches) bind(C, name="ljmd_profile_read") result(status)
      import :: c_int, c_int32_t, c_ptr, c_double
      type(c_ptr), value :: handle
      real(c_double), intent(out) :: ms_avg(4)
      integer(c_int32_t), intent(out) :: launches
      integer(c_int) :: status
    end function
  end interface

contains

  ! Text of the last error (handle = c_null_ptr: last error of a stateless call / failed create).
  function ljmd_error_text(handle) result(text)
    type(c_ptr), intent(in) :: handle
    character(len=:), allocatable :: text
    type(c_ptr) :: p
    character(kind=c_char), pointer :: chars(:)
    integer :: k, n
    p = ljmd_last_error(handle)
    text = ''
    if (.not. c_associated(p)) return
    call c_f_pointer(p, chars, [512])
    n = 0
    do k = 1, 512
      if (chars(k) == c_null_char) exit
      n = k
    end do
    allocate(character(len=n) :: text)
    do k = 1, n
      text(k:k) = chars(k)
    end do
  end function ljmd_error_text

  ! The reference's error convention is `stop 'routine(): message'` (e.g.
  ! lj_potential_energy.f90:77-82): a non-zero status ends the program the same way.
  subroutine ljmd_check(status, handle, where)
    integer(c_int), intent(in) :: status
    type(c_ptr), intent(in) :: handle
    character(len=*), intent(in) :: where
    if (status /= LJMD_OK) then
      write(*, '(a)') 'ljmd: ' // where // ': ' // ljmd_error_text(handle)
      stop 'ljmd: GPU hot path failed'
    end if
  end subroutine ljmd_check

end module ljmd_c_api
